/*
 * mpcqp.h -- C ABI of libmpcqp_hip.so: batched linear-MPC QP build + OSQP-style ADMM solve on
 * AMD MI355X (gfx950).  Plain pointers and sizes only; every function returns 0 on success or a
 * negative mpcqp_error; solver outcomes (solved / infeasible / max-iter ...) are reported in
 * mpcqp_info.status, never as an error code.
 *
 * The entry points mirror, one to one, what the reference's controller does around its solver
 * (file:line relative to the reference checkout):
 *
 *   mpcqp_create          osqp.OSQP()                                   pyMPC/mpc.py:241
 *   mpcqp_setup           MPCController.setup():  _compute_QP_matrices_ pyMPC/mpc.py:254-266,456-608
 *                         + prob.setup(P,q,A,l,u, warm_start=True, eps_abs=, eps_rel=)
 *   mpcqp_update          MPCController.update(): _update_QP_matrices_  pyMPC/mpc.py:338-362,386-454
 *                         + prob.update(l=,u=,q=)
 *   mpcqp_setup_qp        prob.setup(P,q,A,l,u,...) with caller-built vectors     pyMPC/mpc.py:266
 *   mpcqp_update_vectors  prob.update(l=,u=,q=) with caller-built vectors         pyMPC/mpc.py:454
 *   mpcqp_solve           MPCController.solve():  prob.solve()          pyMPC/mpc.py:366-375
 *   mpcqp_get_solution    res.x / res.y / res.info.*                    pyMPC/mpc.py:301-327
 *   mpcqp_get_u0          output(): res.x[(Np+1)nx : (Np+1)nx+nu]        pyMPC/mpc.py:301-304
 *   mpcqp_export_qp       the public attributes P,q,A,l,u               pyMPC/mpc.py:598-606
 *   mpcqp_mpc_step        __controller_function__: update(x,u); output() pyMPC/mpc.py:377-384
 *   mpcqp_mpc_run / _loop the caller loop  u = output(); plant; update() examples/example_point_mass.py:88-101,
 *                         (+ LinearStateEstimator update/predict)        pyMPC/mpc.py:688-692, pyMPC/kalman.py:109-134
 *   mpcqp_share_factor    ONE controller evaluated at many states        test_scripts/example_mpc_function.py:61-64,105-111
 *                         (copies of it share one KKT factor)
 *
 * The QP is the reference's sparse (non-condensed) formulation, bug-for-bug:
 *   w = [x_0..x_Np | u_0..u_{Nc-1} | eps_0..eps_Np],  n = 2(Np+1)nx + Nc nu
 *   rows: dynamics (Np+1)nx | soft state box (Np+1)nx | input box Nc nu | Delta-u (Nc+1)nu
 * All numbers are float64.  A handle owns `batch` independent MPC instances of identical
 * dimensions; per-instance arrays are instance-major and contiguous ([batch][...]).
 * Input/output pointers may be host OR device pointers (copies use hipMemcpyDefault).
 * One handle <-> one HIP stream; calls on one handle must be serialised by the caller.
 */
#ifndef MPCQP_H
#define MPCQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcqp_handle mpcqp_handle;

enum mpcqp_error {
    MPCQP_OK = 0,
    MPCQP_ERR_ARG = -1,          /* bad dimension / NULL pointer / unsupported combination */
    MPCQP_ERR_HIP = -2,          /* a HIP runtime call failed (see mpcqp_last_error)        */
    MPCQP_ERR_NO_DEVICE = -3,    /* no usable GPU                                           */
    MPCQP_ERR_UNSUPPORTED = -4,  /* valid reference configuration not implemented on device */
    MPCQP_ERR_STATE = -5         /* call order (e.g. solve before setup)                    */
};

/* OSQP's status values and strings (what pyMPC compares against 'solved', mpc.py:301,372) */
enum mpcqp_status {
    MPCQP_SOLVED = 1, MPCQP_SOLVED_INACCURATE = 2, MPCQP_MAX_ITER_REACHED = -2,
    MPCQP_PRIMAL_INFEASIBLE = -3, MPCQP_PRIMAL_INFEASIBLE_INACCURATE = 3,
    MPCQP_DUAL_INFEASIBLE = -4, MPCQP_DUAL_INFEASIBLE_INACCURATE = 4,
    MPCQP_NON_CVX = -7, MPCQP_UNSOLVED = -10
};

/* OSQP 0.6.x settings that influence the ADMM iteration (same names, same defaults). */
typedef struct {
    double rho, sigma, alpha;
    double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
    double adaptive_rho_tolerance;
    int32_t max_iter, check_termination, scaling;
    int32_t adaptive_rho, adaptive_rho_interval;   /* interval 0 -> 4*check_termination */
    int32_t warm_start;
    int32_t soft_constraints;   /* pyMPC's SOFT_ON switch (mpc.py:237,530-597): 1 (the only public mode) = state box on x_k + eps_k with slack
                                   columns eps; 0 = hard state box, no slack columns (n = (Np+1)nx + Nc nu).  Fixed at mpcqp_create. */
    int32_t backend;            /* enum mpcqp_backend: which KKT backend the handle runs.  0 = chosen from the shape and the batch (the only
                                   value a caller needs); the others force one -- MPCQP_ERR_UNSUPPORTED if the shape is not eligible for
                                   it.  Fixed at mpcqp_create.  No environment variable is read anywhere in the library. */
    int32_t tuning;             /* bit flags, enum mpcqp_tuning (0 = everything on).  Fixed at mpcqp_create. */
} mpcqp_settings;

/* Values of mpcqp_settings.backend (results do not depend on the backend beyond rounding; tests/test_gpu_backends.py runs the parity
 * suite once per backend a fixture is eligible for). */
enum mpcqp_backend {
    MPCQP_BACKEND_AUTO = 0,     /* from shape and batch, see mpcqp_create */
    MPCQP_BACKEND_SWEEPS = 1,   /* block-tridiagonal sweeps (stages of up to 32) / plain block LDL' (33..128), factor streamed every iteration */
    MPCQP_BACKEND_DENSE = 2,    /* explicit K^-1 in registers: N (nx+nu) <= 128 */
    MPCQP_BACKEND_BCR = 3,      /* block cyclic reduction, factor resident in registers, 256-thread workgroups: nx+nu <= 16, Np <= 30, Nc = Np -- at ANY batch */
    MPCQP_BACKEND_BCR8 = 4,     /* the same with 512-thread workgroups (two waves per SIMD) and the explicit inverse of what two levels of
                                   reduction leave ("dense top").  AUTO picks it for such shapes at EVERY batch when the horizon has 21..30 steps
                                   and nx+nu > 8 (the BASELINE shape (12,4,30)), otherwise up to three instances per compute unit */
    MPCQP_BACKEND_BCRT = 5      /* the dense-top factor on 256-thread workgroups (the comparison BCR8 was measured against) */
};
enum mpcqp_tuning {
    MPCQP_TUNE_NO_BALANCE = 1,  /* keep the identity workgroup -> instance map (no load balancing across compute units) */
    MPCQP_TUNE_NO_LSTAGE = 2,   /* never stage a global-memory iterate into LDS for a round */
    MPCQP_TUNE_NO_GROUPING = 4, /* one stage per 16 x 16 block even where several small stages would share one */
    MPCQP_TUNE_NO_W8 = 8,       /* a single long-horizon controller on grouped stages stays on 256-thread workgroups */
    MPCQP_TUNE_NO_QUEUE = 16,   /* batches beyond the resident workgroup slots: one workgroup per instance (the hardware's dispatch) instead of persistent workgroups taking instances off a queue */
    MPCQP_TUNE_NO_PARTS = 32,   /* persistent closed-loop launches: an instance's whole loop is ONE queue item (no parts) */
    MPCQP_TUNE_QUEUE_SOLVES = 64, /* development: persistent launches for single solves (mpcqp_solve) too, not only for the closed loop */
    MPCQP_TUNE_EVEN_PARTS = 128, /* development: persistent closed-loop launches cut an instance's steps into EQUAL parts (default: decreasing) */
    MPCQP_TUNE_ONE_LAUNCH_SOLVES = 1 << 29, /* development: a solve of more instances than resident slots as ONE persistent launch (instances off the queue, longest expected work first, each run to its end) instead of two launches (one round for everybody, then the unfinished ones re-dealt) */
    MPCQP_TUNE_NO_SHARE = 1 << 30, /* every instance solves with its own factor even where mpcqp_setup finds instances identical to instance 0 (mpcqp_share_factor; measurement switch: results are the same either way) */
    MPCQP_TUNE_SLOTS_SHIFT = 24, /* development: tuning bits 24..28 = resident workgroup slots of a persistent closed-loop launch in eighths of a workgroup per compute unit (8 = one per unit); 0 = the library's choice */
    MPCQP_TUNE_PACE_SHIFT = 8   /* development: tuning bits 8..15 = pacing units (x 3.5 us idled per ADMM iteration by the instances of a fully resident launch of the bandwidth kernels that are not expected to straggle); 0 = off */
};

typedef struct {
    int32_t status;        /* enum mpcqp_status */
    int32_t iter;
    int32_t rho_updates;
    int32_t reserved;
    double obj_val;        /* 1/2 w'Pw + q'w (without the reference's J_CNST)  */
    double pri_res, dua_res;
    double rho;            /* rho in force when the solve ended */
} mpcqp_info;

/* Per-instance controller data, every pointer [batch][...] float64, row-major matrices.
 * Mirrors the constructor arguments of MPCController (mpc.py:76-80); +-inf bounds allowed. */
typedef struct {
    const double *Ad;      /* [nx*nx] */
    const double *Bd;      /* [nx*nu] */
    const double *Qx;      /* [nx*nx] */
    const double *QxN;     /* [nx*nx] */
    const double *Qu;      /* [nu*nu] */
    const double *QDu;     /* [nu*nu] */
    const double *xmin, *xmax;        /* [nx] */
    const double *umin, *umax;        /* [nu] */
    const double *Dumin, *Dumax;      /* [nu] */
    const double *uref;               /* [nu] */
    const double *eps_feas;           /* [1]  */
} mpcqp_model;

void mpcqp_default_settings(mpcqp_settings *s);
const char *mpcqp_status_string(int status);
const char *mpcqp_last_error(void);
int mpcqp_device_count(void);

/* osqp.OSQP() (mpc.py:241) for `batch` controllers of one shape.  Np >= 2, 1 <= Nc <= Np, nx + nu <= 128 (MPCQP_ERR_UNSUPPORTED beyond:
 * the reference has no limit).  The KKT backend is chosen here from the shape and the batch (#CU = compute units of `device`): dense
 * register-resident inverse (N (nx+nu) <= 128, batch <= 6 #CU), cyclic reduction with a dense top on 512-thread workgroups (nx+nu <= 16,
 * Np <= 30, Nc = Np: at every batch for 21 <= Np <= 30 with nx+nu > 8, else batch <= 3 #CU), grouped stages (nx+nu <= 8 on longer horizons),
 * block-tridiagonal sweeps (everything else up to 32 wide), plain block LDL' (33..128 wide).  The cross-over points are measurements on one
 * 256-CU MI355X (LAB_NOTES.md), expressed per compute unit.  Results do not depend on the backend beyond rounding.
 * `s` must come from mpcqp_default_settings (then edited): the struct has no size field, and fields appended by later versions of this header
 * (backend, tuning) are read unconditionally. */
int mpcqp_create(mpcqp_handle **h, int device, int batch, int nx, int nu, int Np, int Nc,
                 const mpcqp_settings *s);
void mpcqp_destroy(mpcqp_handle *h);

/* Run all later work of this handle on an existing HIP stream (hipStream_t passed as void*). */
int mpcqp_set_stream(mpcqp_handle *h, void *hip_stream);
int mpcqp_synchronize(mpcqp_handle *h);

/* setup(): upload the model, build the QP on the device, equilibrate, choose rho, factor the
 * KKT system, cold-start the iterate.  x0 [batch][nx], uminus1 [batch][nu],
 * xref [batch][xref_rows*nx] with xref_rows == 1 (constant reference, mpc.py:497-498) or
 * xref_rows == Np+1 (time-varying reference, mpc.py:489-493). */
int mpcqp_setup(mpcqp_handle *h, const mpcqp_model *model,
                const double *x0, const double *uminus1, const double *xref, int xref_rows);

/* update(): new x0 / u_{-1} / xref (any pointer may be NULL = unchanged); q, l, u are
 * refreshed on the device at the start of the next solve. */
int mpcqp_update(mpcqp_handle *h, const double *x0, const double *uminus1,
                 const double *xref, int xref_rows);

/* The solver seam as pyMPC itself uses it (mpc.py:266 prob.setup(P, q, A, l, u, ...), mpc.py:454 prob.update(l=, u=, q=)):
 * the CALLER builds q, l, u (the reference's own _compute_QP_matrices_ / _update_QP_matrices_) and hands them over
 * verbatim; the library takes q as is and decodes l, u into its bound tables.
 *   mpcqp_setup_qp        the matrices P, A enter through the controller data they are made of (Ad, Bd, Qx, QxN, Qu, QDu,
 *                         eps_feas: pympc_amd.qp_recover reads them back out of the reference-layout P and A and checks
 *                         that rebuilding reproduces both exactly); model->xmin..Dumax are ignored, uref may be NULL;
 *                         q [batch][n], l, u [batch][m] in the reference's layout.
 *   mpcqp_update_vectors  q may be NULL (= unchanged), l and u both or neither (the equality rows l[:nx] == u[:nx] carry x0), like
 *                         osqp.update as pyMPC calls it.  l and u must have the reference's structure (stage-periodic boxes,
 *                         mpc.py:551-580) -- mpcqp_setup_csc and the host wrapper check it.  Host arrays go through a staging block the
 *                         handle keeps; the call is stream-ordered and does not wait.
 * After either call the handle is in "raw vector" mode: solves use these vectors instead of rebuilding q, l, u from
 * (x0, u_{-1}, xref); mpcqp_update / mpcqp_mpc_step / mpcqp_step_host switch back -- on a handle that was set up from raw vectors they
 * must then be given x0, u_{-1} AND xref (it holds none yet: MPCQP_ERR_STATE otherwise) --, mpcqp_mpc_loop refuses (MPCQP_ERR_STATE). */
int mpcqp_setup_qp(mpcqp_handle *h, const mpcqp_model *model, const double *q, const double *l, const double *u);
int mpcqp_update_vectors(mpcqp_handle *h, const double *q, const double *l, const double *u);

/* The same seam with the MATRICES themselves, bindable from any language (mpc.py:266 prob.setup(P, q, A, l, u, ...)):
 *   mpcqp_create_csc  reads the controller's dimensions (nx, nu, Np, Nc) out of the sparsity patterns -- P: upper triangle or full
 *                     symmetric, A: both CSC with int64 column pointers and int32 row indices, shared by the batch -- and creates
 *                     the handle (nx_hint / nu_hint > 0 settle a pattern that does not determine them, e.g. an Ad with an empty first row).
 *                     Slack columns or not (pyMPC's SOFT_ON, mpc.py:237) is read off the pattern too and overrides settings->soft_constraints;
 *   mpcqp_setup_csc   reads (Ad, Bd, Qx, QxN, Qu, QDu, eps_feas) of every instance out of the values P_val [batch][nnz(P)],
 *                     A_val [batch][nnz(A)], REBUILDS both matrices from them as pyMPC/mpc.py:456-608 would and compares entry for
 *                     entry, checks q, l, u ([batch][n], [batch][m], HOST pointers) for pyMPC's structure, and continues as
 *                     mpcqp_setup_qp.  A QP that is not pyMPC's is refused with MPCQP_ERR_UNSUPPORTED (mpcqp_last_error says
 *                     which entries differ); nothing is approximated.  Afterwards: mpcqp_update_vectors / mpcqp_solve / ... */
int mpcqp_create_csc(mpcqp_handle **h, int device, int batch, int n, int m, const int64_t *P_colptr, const int32_t *P_rowidx,
                     const int64_t *A_colptr, const int32_t *A_rowidx, int nx_hint, int nu_hint, const mpcqp_settings *s);
int mpcqp_setup_csc(mpcqp_handle *h, const double *P_val, const double *A_val, const double *q, const double *l, const double *u);

/* Replace the iterate (osqp.warm_start(x=, y=)); x [batch][n], y [batch][m], NULL = keep. */
int mpcqp_warm_start(mpcqp_handle *h, const double *x, const double *y);

/* Change tolerances / iteration limits after setup (osqp.update_settings). */
int mpcqp_update_settings(mpcqp_handle *h, const mpcqp_settings *s);

/* One ADMM solve of every instance: a single kernel launch on the handle's stream (asynchronous -- the
 * mpcqp_get_* calls and mpcqp_synchronize wait for it).  Every instance iterates until ITS termination test
 * passes (checked on the device every check_termination iterations); there is no host loop and no batch-wide
 * barrier between rounds. */
int mpcqp_solve(mpcqp_handle *h);

/* Results of the last solve (synchronises).  x [batch][n], y [batch][m], info [batch];
 * any pointer may be NULL.  u0 [batch][nu] is the first optimal input of each instance; with a DEVICE destination
 * mpcqp_get_u0 / mpcqp_mpc_step are stream-ordered and return without waiting. */
int mpcqp_get_solution(mpcqp_handle *h, double *x, double *y, mpcqp_info *info);
int mpcqp_get_u0(mpcqp_handle *h, double *u0);

/* One control step u = K(x, u_{-1}) (MPCController.__controller_function__, mpc.py:377-384): update(x0, uminus1, xref),
 * warm-started solve, output() -- u_out [batch][nu] receives the first optimal input of every instance, or u_failure
 * (= uref) where the status is not 'solved' (mpc.py:271-336).  Like output(), the call also makes u_out the u_{-1} of
 * the next step, so uminus1 may be NULL from the second step on; xref may be NULL (unchanged).  Synchronises. */
int mpcqp_mpc_step(mpcqp_handle *h, const double *x0, const double *uminus1, const double *xref, int xref_rows,
                   double *u_out);

/* The drop-in class's update() in one call with HOST arrays (mpc.py:338-375: prob.update(l=, u=, q=) + prob.solve() + res.x / res.y /
 * res.info): mpcqp_update(x0, uminus1, xref) (any of them may be NULL = unchanged), a warm-started solve and mpcqp_get_solution.
 * x0 / uminus1 / xref travel in the kernel's arguments (one controller, all three given, at most 32 doubles) or are copied into mapped host
 * memory which the solve kernel reads itself; x [batch][n], y [batch][m],
 * info [batch] (any may be NULL) come back the same way and the call waits on a flag in that memory: one kernel launch, no copy
 * calls, no stream synchronisation -- the latency path of a single small controller.  All pointers are HOST pointers. */
int mpcqp_step_host(mpcqp_handle *h, const double *x0, const double *uminus1, const double *xref, int xref_rows,
                    double *x, double *y, mpcqp_info *info);

/* Device-side receding-horizon loop: K closed-loop steps of every instance without host round trips,
 *     for k in range(nsteps):  u = K.output();  x = Ap x + Bp u + w[k];  K.update(x)       (solve included)
 * i.e. the caller loop of examples/example_point_mass.py:88-101 / pyMPC/mpc.py:688-692 with a linear plant.
 * Needs a solved handle (mpcqp_setup + mpcqp_solve).  output() follows mpc.py:271-336: the first input of the
 * last solution if its status is 'solved', else u_failure (= uref).  The reference xref stays as last uploaded.
 *   w           [nsteps][batch][nx] additive disturbance, or NULL
 *   Ap, Bp      [batch][nx*nx], [batch][nx*nu] plant matrices, or NULL (plant = the controller's Ad, Bd)
 *   x_traj      [nsteps+1][batch][nx] states x_0..x_K (x_0 = the state of the last update/setup), or NULL
 *   u_traj      [nsteps][batch][nu] applied inputs, or NULL
 *   status_traj, iter_traj  [nsteps][batch] status / ADMM iterations of the solve after step k's update, or NULL
 * Host or device pointers.  Buffers in device memory are read and written in place by the kernel and the call is then
 * stream-ordered (it returns without waiting, like mpcqp_get_u0 with a device destination); with any host output the call
 * returns when the run is complete.  Afterwards the handle holds the solution for x_K (mpcqp_get_solution /
 * mpcqp_get_u0 work as after mpcqp_solve). */
int mpcqp_mpc_run(mpcqp_handle *h, int nsteps, const double *w, const double *Ap, const double *Bp,
                  double *x_traj, double *u_traj, int32_t *status_traj, int32_t *iter_traj);

/* The same loop with everything optional spelled out (zero-initialise, then set what is used):
 *  - a time-varying reference: the solve after step k uses xref_traj[k] (mpc.py:338-364 update(x, u, xref));
 *  - output feedback: the controller does not see the plant state but the estimate of a LinearStateEstimator
 *    (pyMPC/kalman.py:109-134; model = the controller's Ad, Bd), as in examples/example_inverted_pendulum_kalman.py:135-174:
 *        y = C x + v[k];  u = K.output();  x = Ap x + Bp u + w[k];  KF.update(y);  KF.predict(u);  K.update(KF.x, u)
 *    The estimate at step 0 is the state of the last update/setup; the true plant state is x_true (in/out). */
typedef struct {
    const double *w;            /* [nsteps][batch][nx] process disturbance, or NULL */
    const double *Ap, *Bp;      /* [batch][nx*nx], [batch][nx*nu] plant, or NULL (the controller's Ad, Bd) */
    const double *xref_traj;    /* [nsteps][batch][xref_rows*nx], or NULL */
    int32_t ny;                 /* > 0 switches output feedback on */
    int32_t xref_rows;          /* rows of one xref_traj entry: 1 or Np+1 (like mpcqp_update: the reference shape may change,
                                   mpc.py:414-424); 0 = the shape of the last upload.  Anything else: MPCQP_ERR_ARG */
    const double *C;            /* [batch][ny*nx] */
    const double *Lgain;        /* [batch][nx*ny] Kalman (filter) gain */
    const double *v;            /* [nsteps][batch][ny] measurement noise, or NULL */
    double *x_true;             /* [batch][nx] true plant state: in = at step 0, out = after the run */
    double *x_traj;             /* [nsteps+1][batch][nx] plant states, or NULL */
    double *xhat_traj;          /* [nsteps+1][batch][nx] estimates xhat[k|k-1] (output feedback only), or NULL */
    double *y_traj;             /* [nsteps][batch][ny] measurements (output feedback only), or NULL */
    double *u_traj;             /* [nsteps][batch][nu], or NULL */
    int32_t *status_traj, *iter_traj;   /* [nsteps][batch], or NULL */
} mpcqp_loop;
int mpcqp_mpc_loop(mpcqp_handle *h, int nsteps, const mpcqp_loop *io);

/* Cumulative work counters since creation / last reset: out4 = { ADMM iterations, residual
 * evaluations, refactorizations, instance-solves } summed over the batch (synchronises). */
int mpcqp_get_stats(mpcqp_handle *h, uint64_t *out4, int reset);

/* Timing of the solve kernel (k_mpc_run: mpcqp_solve / mpcqp_iterate / mpcqp_mpc_run launches), measured with
 * HIP events on the handle's stream: enable = 1/0 to switch, -1 to leave unchanged; returns accumulated
 * milliseconds and launch count (synchronises on the pending launches), optionally resets. */
int mpcqp_profile(mpcqp_handle *h, int enable, double *run_ms, int64_t *run_launches, int reset);

/* When each instance's workgroup entered and left the last mpcqp_mpc_loop / mpcqp_mpc_run launch and when it finished each of its first
 * nsteps closed-loop steps (0 <= nsteps <= 64): out [batch][2 + nsteps] = { entry, exit, end of step 0, end of step 1, ... } in ticks of the
 * GPU's constant 100 MHz clock (s_memrealtime: one time base for the whole chip).  A launch ends with its slowest instance; this is what says
 * how much work was done while the compute units were still full (bench.py: roofline.frac_excluding_tail).  Launches of fewer than 64 steps: slot
 * [2 + 63] holds (XCC_ID << 32 | HW_ID) of the workgroup that ran the instance's first steps (scripts/diag_makespan.py).  Synchronises. */
int mpcqp_get_launch_times(mpcqp_handle *h, uint64_t *out, int nsteps);

/* The controller's dimensions (what mpcqp_create was given, or what mpcqp_create_csc read out of the patterns). */
int mpcqp_get_shape(mpcqp_handle *h, int *nx, int *nu, int *Np, int *Nc);

/* Problem sizes: n, m of one instance, bytes of the KKT factor per instance, and nnz(L). */
int mpcqp_get_dims(mpcqp_handle *h, int *n, int *m, int64_t *factor_doubles, int64_t *nnzL);

/* Bytes one instance streams between memory and its compute unit by design of this implementation: per ADMM iteration
 * (the KKT factor: forward blocks twice, S^-1 once; plus the iterate where it does not fit LDS), per round of
 * check_termination iterations (residual evaluation inputs, iterate in/out of LDS), per solve (QP refresh, write-out).
 * The numerator of the HBM roofline bench.py reports (counterpart of the per-iteration cost 2 nnz(L) + O(n+m) of the
 * sparse LDL' solve behind pyMPC/mpc.py:369). */
int mpcqp_get_stream_bytes(mpcqp_handle *h, int64_t *per_iter, int64_t *per_round, int64_t *per_solve);

/* Matrix-core instructions (v_mfma_f64_4x4x4_4b_f64: 512 flop each) one instance issues per ADMM iteration with this handle's backend. */
int mpcqp_get_work(mpcqp_handle *h, int64_t *mfma_per_iter);

/* Workgroups of the handle's solve kernel one compute unit holds at a time, the compute units of its device, threads per workgroup
 * (bench.py: instances in flight = the product of the first two -- the working set the memory-side cache sees). */
int mpcqp_get_occupancy(mpcqp_handle *h, int *workgroups_per_cu, int *compute_units, int *threads_per_workgroup);

/* Name of the solve-kernel instantiation this handle launches (loop = 0: mpcqp_solve; 1: mpcqp_mpc_loop), as profilers print it. */
int mpcqp_kernel_name(mpcqp_handle *h, int loop, char *buf, int buflen);

/* ---- verification surface (used by the parity tests) ------------------------------------ */
/* Materialise what the device built: dense row-major P [batch][n*n], A [batch][m*n],
 * q [batch][n], l,u [batch][m] (the reference's public attributes, mpc.py:598-606). */
int mpcqp_export_qp(mpcqp_handle *h, double *P, double *A, double *q, double *l, double *u);
/* Equilibration D [batch][n], E [batch][m], c [batch]; current rho [batch]. */
int mpcqp_get_scaling(mpcqp_handle *h, double *D, double *E, double *c, double *rho);
/* Solve the (reduced) KKT system of the current factor: K sol = rhs, [batch][n] each. */
int mpcqp_debug_kkt_solve(mpcqp_handle *h, const double *rhs, double *sol);
/* Current ADMM iterate in unscaled units: x [batch][n], z,y [batch][m]. */
int mpcqp_get_iterate(mpcqp_handle *h, double *x, double *z, double *y);
/* Run exactly `iters` ADMM iterations with no termination test and no rho adaptation. */
int mpcqp_iterate(mpcqp_handle *h, int iters);
/* Recompute the KKT factor of every instance from its current rho -- the work one adaptive-rho update costs inside
 * prob.solve() (mpc.py:369; OSQP refactors the KKT matrix whenever it changes rho).  The factor written is the one already in
 * place: nothing observable changes.  Asynchronous on the handle's stream; bench.py times it to report what a refactorization
 * costs per instance. */
int mpcqp_refactor(mpcqp_handle *h);
/* One model, many states -- the caller of test_scripts/example_mpc_function.py:105-111 (10 000 random (x, u_{-1}) through ONE controller) and SURVEY 8(e)'s
 * last paragraph (broadcast the model, scatter only x0): every instance whose factorization inputs (model, rho vector, scaling, cost scale) are bit-identical
 * to instance 0's solves with ONE shared copy of instance 0's factor instead of its own -- the streaming backends then read the factor out of L2 instead of
 * HBM.  Every setup call does this by itself (one map kernel; mpcqp_settings.tuning & MPCQP_TUNE_NO_SHARE: not): a batch set up with the same model and the
 * same x0 / u_{-1} / xref in every instance -- what ONE reference controller's setup() is -- shares from the start, and mpcqp_update then scatters the states.
 * Results are bit-identical to the unshared batch: the factorization is deterministic, and an instance that refactors later (a rho update, changed
 * constraint types, mpcqp_refactor) writes its own slot and solves with that from then on.  This call rebuilds the map against instance 0's factor as of
 * NOW (e.g. after the whole batch has adapted rho the same way) and reports it.  *nshared (may be NULL; non-NULL makes the call synchronous): instances
 * sharing, 0 for the register-resident backends (MPCQP_BACKEND_DENSE / BCR*: they read their factor once per launch, there is nothing to share). */
int mpcqp_share_factor(mpcqp_handle *h, int *nshared);

/* The EQUALITY-constrained part of the handle's QP -- minimise 1/2 w'P w + q'w subject to the dynamics rows alone, every other row
 * ignored -- by sweeps of the method of multipliers in residual form, each one KKT solve with the handle's factor (one factorization,
 * made by mpcqp_setup; give the handle a large rho: a sweep contracts the error by about |P| / (1e3 rho)).  This is what the gains of
 * the law without inequality constraints need (test_scripts/alternative/unconstrained.py:170-183: k_x0, k_Xref, k_Uref, k_uminus1 --
 * there a dense condensed solve); pympc_amd/unconstrained.py calls it on a batch of unit-vector problems.
 * At most `sweeps` sweeps; an instance stops earlier once a correction is below tol * max(1, |w|_inf) (tol = 0: never).
 * cold != 0: start from zero, else continue from the current iterate.  The result is read with mpcqp_get_solution (status 'solved', or
 * 'maximum iterations reached' if a tolerance was given and `sweeps` sweeps did not settle the instance; iter = sweeps done).  res (host or device, may be NULL): [batch][5] = |P w + q + A_e'y|, max(|P w|, |A_e'y|, |q|), |A_e w - b|,
 * max(|A_e w|, |b|) (infinity norms, unscaled) after the last sweep, and the sweeps done.  Synchronous. */
int mpcqp_eq_solve(mpcqp_handle *h, int sweeps, int cold, double tol, double *res);

#ifdef __cplusplus
}
#endif
#endif

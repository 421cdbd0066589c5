// mpcqp_run.h -- part of libmpcqp_hip (included by mpcqp.hip and mpcqp_w8.hip, one translation unit each, behind mpcqp_phases.h).
// The arguments of the persistent solve / closed-loop kernel k_mpc_run (mpcqp_kernels.h) and how a device function reaches them: the phases
// of a solve -- and the latency round, which runs OSQP's termination test itself (mpcqp_latw.h) -- re-read them from the kernel-argument segment.
#pragma once

struct RunArgs {
    int nsteps;                   // closed-loop steps (LOOP kernels); 0 = one solve of the current data (mpcqp_solve)
    int plain;                    // run exactly max_iter iterations, no termination test / rho adaptation (mpcqp_iterate)
    int warm_x;                   // x was replaced by mpcqp_warm_start: begin with z = A x
    int part;                     // LOOP = false only.  0: the whole solve.  1: begin + first round; instances that are not
                                  // finished are appended to `pending`.  2: continue the `pending` instances to the end
                                  // (the launch that follows part 1: its workgroup -> instance map IS the pending list, so
                                  // the unfinished instances spread evenly over the CUs instead of staying where they were)
                                  // 3: no solve at all -- refactor every instance with its current rho (mpcqp_refactor)
    int *pending, *npending;      // [batch] instance list and its length (device)
    int max_iter, chk, rho_every;
    const double *w;              // [nsteps][batch][nx] additive plant disturbance, or null
    const double *Ap, *Bp;        // [batch][nx*nx], [batch][nx*nu] plant matrices, or null (plant = model Ad, Bd)
    const double *xref_traj;      // [nsteps][batch][xref_blk] reference for the solve after step k, or null (unchanged)
    int xref_blk;                 // xref_rows * nx
    int ny;                       // > 0: output feedback through a LinearStateEstimator (pyMPC/kalman.py:109-134)
    const double *C, *Lg, *v;     // [batch][ny*nx], [batch][nx*ny], [nsteps][batch][ny] (or null)
    double *x_true;               // [batch][nx] true plant state (in/out) when the controller only sees the estimate
    double *x_traj;               // [nsteps+1][batch][nx] plant states
    double *xhat_traj;            // [nsteps+1][batch][nx] estimates xhat[k|k-1] handed to update() (estimator only)
    double *y_traj;               // [nsteps][batch][ny] measurements (estimator only)
    double *u_traj;               // [nsteps][batch][nu]
    int *status_traj, *iter_traj; // [nsteps][batch]: outcome of the solve that follows step k's update
    int batch;
    // persistent launch (k_mpc_run: the grid is the resident slots, workgroups take instances off a queue); null = one workgroup per instance
    int *vcur;                    // [grid] the map entry (instance | pace bits) each workgroup is working on: what Ptrs::perm points at
    const int *vperm;             // [batch] queue position -> map entry, or null = identity
    unsigned *vqueue;             // next queue position (zeroed before the launch)
    int vparts, voff[17];         // closed loop: queue items per instance (0 / 1: the whole loop is one item) and their first steps (voff[p] .. voff[p + 1])
    int *vdone;                   // [batch] steps of this launch an instance has completed (zeroed before the launch; vparts > 1)
    // host-resident exchange (mpcqp_step_host: one launch per control step, no copy calls, no stream synchronisation):
    const double *pin_in;         // [batch][pin_stride] = [x0 | u_{-1} | xref] in mapped host memory, copied into the step blob first (null = off)
    int pin_stride, pin_mask, pin_xref;   // mask: 1 x0, 2 u_{-1}, 4 xref (pin_xref doubles)
    double *pub;                  // mapped host memory: [batch][n] x | [batch][m] y | [batch] info | flag; written when the solve is done (null = off)
    unsigned *done;               // device counter of finished workgroups (the last one raises the flag)
    unsigned long long seq;       // value the flag takes
    // ... or, a single controller whose [x0 | u_{-1} | xref] is a few doubles: in the kernel arguments themselves (no PCIe read in front of the solve)
    int inl_n;                    // doubles in inl (0 = off)
    double inl[32];
};

__host__ __device__ inline int next_stop(int iter, int max_iter, int chk, int rho_every) {
    int nxt = max_iter;
    if (chk) { int v = (iter / chk + 1) * chk; nxt = v < nxt ? v : nxt; }
    if (rho_every) { int v = (iter / rho_every + 1) * rho_every; nxt = v < nxt ? v : nxt; }
    return nxt;
}
__host__ __device__ inline int stop_mode(int iter, int max_iter, int chk, int rho_every, bool plain) {
    int mode = plain ? COLD_PLAIN : 0;
    if (chk && iter % chk == 0) mode |= COLD_CHECK;
    if (rho_every && iter % rho_every == 0) mode |= COLD_RHO;
    if (iter == max_iter && !plain) mode |= COLD_FINAL;
    return mode;
}

// The three phases are separate (non-inlined) functions so that each gets a register allocation of its own --
// inlined into one body, the cold code's live ranges pushed spill reloads into the ADMM sweep.  They take no
// pointer arguments: everything is re-read from the kernel-argument segment, which is uniform, constant memory
// (scalar loads), instead of travelling through the vector-register calling convention.
struct RunKArgs { Lay L; Ptrs P; mpcqp_settings S; RunArgs R; };
static_assert(sizeof(RunKArgs) % 8 == 0, "hidden kernel arguments start right behind RunKArgs");
typedef const __attribute__((address_space(4))) RunKArgs *ckargs;
// (In a non-kernel function the kernarg segment pointer itself is not available, the implicit-argument pointer is:
//  the hidden arguments follow the explicit ones, here the single RunKArgs struct, at the next 8-byte boundary.)
__device__ __forceinline__ const RunKArgs &run_kargs() {
    typedef const __attribute__((address_space(4))) char *cbytes;
    return *(const RunKArgs *)(ckargs)((cbytes)__builtin_amdgcn_implicitarg_ptr() - ((sizeof(RunKArgs) + 7) & ~size_t(7)));
}

// Every phase takes `frame_pin` (see run_admm_phase below: what keeps its calls from being marked `tail`, and with that the phase free of the
// calling convention's callee-saved set); FRAME_PIN is the caller's side of it.
#define PHASE_PIN_USE(p) asm volatile("" :: "v"(p) : "memory")
struct FramePin { int v; __device__ __forceinline__ FramePin() : v(0) { asm volatile("" : "+v"(v)); } __device__ __forceinline__ ~FramePin() { asm volatile("" :: "v"(v)); } };

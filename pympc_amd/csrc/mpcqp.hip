// mpcqp.hip -- hand-written HIP (gfx950 / MI355X) implementation of pyMPC's QP hot path:
//   * construction of the MPC quadratic program from (Ad,Bd,Qx,QxN,Qu,QDu,bounds)   [mpc.py:386-608]
//   * OSQP-style ADMM solve of that program, one persistent workgroup per MPC instance [mpc.py:266,369,454]
//
// Design:
//   - the QP matrices are never stored; P, A, A' are applied matrix-free from the stage data
//     (row "visitors" below enumerate the nonzeros of one row exactly as the reference lays them out);
//   - the ADMM linear system is the reduced KKT matrix  K = c P + sigma D^-2 + A' diag(rho E^2) A
//     (OSQP's quasi-definite KKT with the constraint block eliminated, expressed in UNSCALED variables:
//     Ruiz scaling D,E,c enters only through the metric vectors  s = sigma/D^2  and  omega = rho E^2);
//     with the slack variables eliminated it is block tridiagonal along the horizon with (nx+nu)^2 blocks
//     and is factored by a block LDL' whose factor, stored in FP64-MFMA operand order, streams from HBM/L2
//     every iteration;
//   - one 256-thread workgroup owns one instance for the whole solve (or K closed-loop steps): iterate in LDS,
//     two waves run the sequential block forward/backward half-sweeps of the twisted factorization on the matrix
//     cores (v_mfma_f64_4x4x4_4b_f64, stage output registers = next stage's B operand), all waves run the
//     stage-parallel parts.
//
//   - that is the bandwidth backend (any size, large batches).  Further KKT backends, chosen per handle at mpcqp_create (DESIGN.md
//     section 3): block cyclic reduction with the factor resident on the compute unit for 16 x 16 stages at small batches -- on 512-thread
//     workgroups with a dense top (mpcqp_latw.h, compiled in mpcqp_w8.hip; mpcqp_bcr.h, mpcqp_lat.h: the 256-thread original) --, an explicit K^-1
//     held in registers for the reference's own small examples (mpcqp_dense.h, mpcqp_tiny.h), grouped small stages for long horizons
//     (mpcqp_group.h), and plain vector-ALU block LDL' for stages wider than 32 (mpcqp_wide.h, mpcqp_huge.h).
//
// This file: host side and C ABI (include/mpcqp.h).  Device code: mpcqp_layout.h, mpcqp_qp.h, mpcqp_factor.h, mpcqp_sweeps.h,
// mpcqp_group.h, mpcqp_bcr.h, mpcqp_wide.h, mpcqp_huge.h, mpcqp_dense.h, mpcqp_border.h, mpcqp_phases.h, mpcqp_tiny.h, mpcqp_lat.h, mpcqp_latw.h,
// mpcqp_kernels.h (the last block also compiled at 512 threads per workgroup by mpcqp_w8.hip); host only:
// mpcqp_csc.h (one translation unit).
//
// FP64 throughout.  No CPU fallback exists in this library.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"

#include "mpcqp_defs.h"

// one polite spin of a host-side busy wait
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(MPCQP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

#include "mpcqp_layout.h"
#include "mpcqp_qp.h"
#include "mpcqp_factor.h"
#include "mpcqp_sweeps.h"
#include "mpcqp_group.h"
#include "mpcqp_bcr.h"
#include "mpcqp_wide.h"
#include "mpcqp_huge.h"
#include "mpcqp_dense.h"
#include "mpcqp_border.h"
#include "mpcqp_phases.h"
#include "mpcqp_run.h"
#include "mpcqp_tiny.h"
#include "mpcqp_lat.h"
#include "mpcqp_latw.h"
#include "mpcqp_kernels.h"
#include "mpcqp_latw_check.h"
#include "mpcqp_csc.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int BCR_STAGES = 31;      // the largest stage count the register-resident cyclic reduction is instantiated for (Np = 30: the BASELINE shape)
// The static schedule of mpcqp_lat.h exists for 11, 21 and 31 stages; a problem runs the smallest one that holds its N = Np + 1 stages (the rest
// are identity padding: factor_bcr).  0: too long a horizon for this backend.
static int bcr_schedule(int N) { return N <= 11 ? 11 : N <= 21 ? 21 : N <= BCR_STAGES ? BCR_STAGES : 0; }
// AUTO at up to one instance per compute unit: the eight-wave kernels of mpcqp_w8.hip (measured against the four-wave ones: DESIGN.md section 5b)
#ifndef MPCQP_AUTO_BCR8
#define MPCQP_AUTO_BCR8 1
#endif
// mpcqp_w8.hip: the second translation unit (NT = 512).  kargs: this unit's RunKArgs, byte for byte the other's.
int mpcqp_w8_launch(const void *kargs, size_t kargs_bytes, int spec12_4, int sched, int loop, int grid, size_t smem, hipStream_t stream);
#ifdef MPCQP_RUN_TIMING
void mpcqp_w8_ticks(unsigned long long *out16);
#endif
constexpr int BALANCE_EVERY = 16;  // solves between two rebuilds of the workgroup -> instance map
constexpr int QUEUE_ITEMS_PER_SLOT = 16;   // persistent closed-loop launches: queue items per resident workgroup slot (run_kernel_args)
constexpr int BALANCE_FIRST = 4;   // ... before the first one (an instance shows its character within a few solves; a short run should not end unbalanced)

struct PackItem { const double *src; double *dst; int w, stride; };
struct PackArgs { PackItem it[24]; int n, batch; };
struct mpcqp_handle {
    int device, batch;
    Lay L;
    Ptrs P;
    mpcqp_settings S;
    hipStream_t stream;
    bool own_stream, is_setup, lds_state;
    size_t smem_setup, smem_solve;
    std::vector<void *> allocs;
    double *u0_dev;
    int *pending_dev, *npending_dev;   // two-launch solve: instances that need more than the first round
    int *perm_dev;                // [batch] workgroup -> instance map (P.perm points here once a map has been built)
    int *qperm_dev; bool qperm_set;     // [batch] the queue order of persistent launches (longest expected work first), once a map has been built
    int *vcur_dev, *vdone_dev; unsigned *vqueue_dev;   // persistent launches: [slots] map entry each workgroup is working on; [batch] closed-loop steps done; the queue position
    std::vector<double> work_ema; // per instance: smoothed ADMM iterations per balancing interval (host)
    int ncu, solves_since_balance, auto_balance;
    void *run_buf; size_t run_bytes;     // staging of mpcqp_mpc_run (disturbances, plant, trajectories)
    bool profiling;
    hipEvent_t ev0[MAXEV], ev1[MAXEV];   // ring of event pairs around the solve-kernel launches
    long long ev_count;
    double run_ms;
    long long run_launches;
    int nevents;                         // event pairs created so far (a partially built handle is destroyed cleanly)
    bool warm_x_pending;                 // mpcqp_warm_start replaced x: the next solve starts from z = A x (osqp_warm_start)
    // mpcqp_step_host: mapped, coherent host memory the kernel reads its step data from and writes its results to
    double *pin_in, *pin_out; void *pin_in_dev, *pin_out_dev;
    unsigned *done_dev; unsigned long long host_seq; int pin_stride; bool pin_tried;
    PackArgs pack;                       // device-resident sources of put() waiting for flush_puts
    CscSeam *csc;                        // patterns of P and A of a handle made by mpcqp_create_csc
    double *vec_buf;                     // staging of mpcqp_update_vectors' host arrays [q | l | u], allocated on first use, kept
    bool step_blank;                     // set up through mpcqp_setup_qp: the step blob holds no x0 / u_{-1} / xref yet
    int *fown_dev; unsigned *nshared_dev;   // mpcqp_share_factor: [batch] factor slot per instance (P.fown points here while sharing is on); how many share
};

extern "C" void mpcqp_default_settings(mpcqp_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 4000; s->check_termination = 25; s->scaling = 10;
    s->adaptive_rho = 1; s->adaptive_rho_interval = 0; s->warm_start = 1; s->soft_constraints = 1;
    s->backend = MPCQP_BACKEND_AUTO; s->tuning = 0;
}

extern "C" const char *mpcqp_status_string(int status) {
    switch (status) {
        case MPCQP_SOLVED: return "solved";
        case MPCQP_SOLVED_INACCURATE: return "solved inaccurate";
        case MPCQP_MAX_ITER_REACHED: return "maximum iterations reached";
        case MPCQP_PRIMAL_INFEASIBLE: return "primal infeasible";
        case MPCQP_PRIMAL_INFEASIBLE_INACCURATE: return "primal infeasible inaccurate";
        case MPCQP_DUAL_INFEASIBLE: return "dual infeasible";
        case MPCQP_DUAL_INFEASIBLE_INACCURATE: return "dual infeasible inaccurate";
        case MPCQP_NON_CVX: return "problem non convex";
        default: return "unsolved";
    }
}

extern "C" const char *mpcqp_last_error(void) { return g_err.c_str(); }

extern "C" int mpcqp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static Lay make_layout(int nx, int nu, int Np, int Nc, int soft) {
    Lay L; memset(&L, 0, sizeof(L));
    L.nx = nx; L.nu = nu; L.Np = Np; L.Nc = Nc; L.N = Np + 1; L.nb = nx + nu;
    L.n_x = L.N * nx; L.n_u = Nc * nu;
    L.soft = soft ? 1 : 0;
    L.n = (L.soft ? 2 : 1) * L.n_x + L.n_u; L.m = 2 * L.n_x + L.n_u + (Nc + 1) * nu;
    L.ou = L.n_x; L.oe = L.n_x + L.n_u;
    L.rs = L.n_x; L.ri = 2 * L.n_x; L.rdu = 2 * L.n_x + L.n_u;
    L.NB = L.nb <= 16 ? 16 : L.nb <= 32 ? 32 : L.nb <= 64 ? 64 : 128;
    L.border = Nc < Np ? 1 : 0;
    L.NcT = L.border ? Nc - 1 : Nc;
    L.rnx = 1.0f / (float)nx; L.rnu = 1.0f / (float)nu;
    int o = 0;
    L.oAd = o; o += nx * nx; L.oBd = o; o += nx * nu;
    L.oxmin = o; o += nx; L.oxmax = o; o += nx;
    L.oumin = o; o += nu; L.oumax = o; o += nu; L.oDumin = o; o += nu; L.oDumax = o; o += nu;
    L.ouref = o; o += nu; L.oeps = o; o += 1;
    L.hot_sz = o;
    L.oQx = o; o += nx * nx; L.oQxN = o; o += nx * nx; L.oQu = o; o += nu * nu; L.oQDu = o; o += nu * nu;
    L.model_sz = o;
    L.hot_lds = L.hot_sz;                              // (mpcqp_create: model_sz where the weight matrices fit in LDS beside everything else)
    L.odu0 = nx + nu + L.N * nx;
    L.step_sz = L.odu0 + 2 * nu;
    L.raw = 0;
    L.xref_rows = 1;
    L.fstage = L.NB == 16 ? FactorFmt<16>::STAGE : L.NB == 32 ? FactorFmt<32>::STAGE : L.NB == 64 ? WideFmt::STAGE : HugeFmt::STAGE;
    L.fhead = L.NB == 16 ? FactorFmt<16>::HEAD : L.NB == 32 ? FactorFmt<32>::HEAD : 0;
    L.ffwd = L.NB == 16 ? FactorFmt<16>::FWD : L.NB == 32 ? FactorFmt<32>::FWD : L.NB == 64 ? WideFmt::NN : HugeFmt::NN;      // (non-zero: the factor starts at the stages)
    L.ftab = L.NB == 16 ? FactorFmt<16>::TAB : L.NB == 32 ? FactorFmt<32>::TAB : 0;
    L.tsz = L.m + L.N * L.NB * (L.NB >= 64 ? 2 : 1);   // [W (m) | Tc (N*NB)] (+ the second stage-major vector of the wide solve); mpcqp_create widens it where the factorization needs more
    L.NR = L.N * L.nb; L.dld = L.NR | 1;               // dense mode (decided in mpcqp_create): unknowns, odd LDS row stride
    L.nw = NWAVES; L.bcrtop = 0;                        // (mpcqp_create: eight waves and a dense top for the kernels of mpcqp_w8.hip)
    return L;
}

template <class T>
static int dalloc(mpcqp_handle *h, T **p, size_t count) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, count * sizeof(T) + 16));
    HIPCHK(hipMemsetAsync(q, 0, count * sizeof(T) + 16, h->stream));
    h->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

extern "C" int mpcqp_create(mpcqp_handle **out, int device, int batch, int nx, int nu, int Np, int Nc, const mpcqp_settings *s) {
    if (!out || batch < 1 || nx < 1 || nu < 1 || Np < 2 || Nc < 1 || Nc > Np) return fail(MPCQP_ERR_ARG, "mpcqp_create: bad dimensions");
    if (nx + nu > 128) return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_create: nx+nu > 128 is not implemented");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPCQP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MPCQP_ERR_ARG, "mpcqp_create: bad device index");
    HIPCHK(hipSetDevice(device));
    mpcqp_handle *h = new mpcqp_handle();
    h->device = device; h->batch = batch; h->is_setup = false; h->u0_dev = nullptr; h->run_buf = nullptr; h->run_bytes = 0; h->perm_dev = nullptr; h->vcur_dev = nullptr; h->vdone_dev = nullptr; h->vqueue_dev = nullptr; h->qperm_dev = nullptr; h->qperm_set = false; h->solves_since_balance = 0; h->auto_balance = 1; h->ncu = 0;
    h->profiling = false; h->run_ms = 0.0; h->run_launches = 0; h->ev_count = 0; h->nevents = 0; h->stream = nullptr; h->own_stream = false;
    h->warm_x_pending = false;
    h->csc = nullptr; h->vec_buf = nullptr; h->step_blank = false;
    h->pin_in = h->pin_out = nullptr; h->pin_in_dev = h->pin_out_dev = nullptr; h->done_dev = nullptr; h->host_seq = 0; h->pin_stride = 0; h->pin_tried = false;
    if (s) h->S = *s; else mpcqp_default_settings(&h->S);
    h->L = make_layout(nx, nu, Np, Nc, h->S.soft_constraints);
    // (every failure from here on releases what has been created so far: mpcqp_destroy copes with a partial handle)
    auto hip_fail = [&](hipError_t e, const char *what) { std::string msg = std::string(what) + ": " + hipGetErrorString(e); mpcqp_destroy(h); return fail(MPCQP_ERR_HIP, msg); };
    { hipError_t e = hipStreamCreate(&h->stream); if (e != hipSuccess) { h->stream = nullptr; return hip_fail(e, "hipStreamCreate"); } }
    h->own_stream = true;
    for (int i = 0; i < MAXEV; ++i) {
        hipError_t e = hipEventCreate(&h->ev0[i]); if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
        e = hipEventCreate(&h->ev1[i]); if (e != hipSuccess) { hipEventDestroy(h->ev0[i]); return hip_fail(e, "hipEventCreate"); }
        h->nevents = i + 1;
    }
    const Lay &L = h->L;
    Ptrs &P = h->P; memset(&P, 0, sizeof(P));
    size_t B = (size_t)batch;
    // Small problems keep the iterate x, z, y in LDS behind the common block (four workgroups per CU: 40 KB each); larger
    // ones keep it in L2/HBM.
    const size_t state_doubles = (size_t)(L.n + 2 * L.m);
    // (an LDS-resident iterate's termination check reads the weight matrices from LDS too -- check_norms_own: behind W in the work area unless
    //  they are staged with the hot prefix -- so the work area of such a handle holds at least W and them: short horizons of wide stages only)
    const int tsz_plain = L.tsz;
    if (L.NB <= 32) h->L.tsz = std::max(L.tsz, L.m + (L.model_sz - L.hot_sz));
    h->lds_state = L.NB <= 32 && sizeof(double) * ((size_t)smem_common_doubles(L) + state_doubles) <= 40 * 1024 && L.m <= 4 * NT && L.N * L.NB <= 2 * NT && L.n_u + L.nu <= NT;      // (the owner map of the parallel phases: two state elements and one input element per thread)
    if (!h->lds_state) h->L.tsz = tsz_plain;
    // The smallest ones (the reference's own examples) solve the KKT system with a register-resident dense inverse (mpcqp_dense.h).
    // (mpcqp_settings.backend forces a choice -- what tests/test_gpu_backends.py runs every eligible fixture through; a forced backend the shape
    //  is not eligible for is refused, never silently replaced)
    const int want = h->S.backend;
    auto refuse = [&](const char *what) { mpcqp_destroy(h); return fail(MPCQP_ERR_UNSUPPORTED, std::string("mpcqp_create: backend ") + what + " is not available for this shape"); };
    if (want < MPCQP_BACKEND_AUTO || want > MPCQP_BACKEND_BCRT) { mpcqp_destroy(h); return fail(MPCQP_ERR_ARG, "mpcqp_create: unknown mpcqp_settings.backend"); }
    const bool dense_shape = h->lds_state && L.NB == 16 && L.NR <= DenseFmt::ROWS;      // (Nc < Np included: the dense inverse holds the held input's couplings itself)
    if (want == MPCQP_BACKEND_DENSE && !dense_shape) return refuse("DENSE");
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->ncu = prop.multiProcessorCount; }
    // (AUTO: up to six instances per compute unit -- measured on (3,1,30), (4,1,20), (2,2,12), (6,2,10), (12,4,7), device loop: ahead of both other backends by 1.2 - 2.3 x up to
    //  512 instances, by 0.94 - 1.4 x at 1024, behind the bandwidth kernel by 0 - 18 % at 2048)
    const bool dense = dense_shape && (want == MPCQP_BACKEND_DENSE || (want == MPCQP_BACKEND_AUTO && (h->ncu <= 0 || batch <= 6 * h->ncu)));
    h->L.dense = dense ? 1 : 0;
    if (dense) h->L.tsz += DenseFmt::SCRATCH;
    // Up to three instances per compute unit (one resident, the others queued): the latency backend (block cyclic reduction, factor resident on the compute unit) for
    // 16 x 16 stages and horizons of up to 30 steps -- the BASELINE shape (12, 4, 30) with compile-time dimensions, anything else with nx + nu <= 16 through
    // the generic instantiations.  Larger batches stream the chain format (the bandwidth backend).
    const bool bcr_shape = !dense && h->lds_state && L.NB == 16 && !L.border && bcr_schedule(L.N) > 0;
    const bool want_bcr = want == MPCQP_BACKEND_BCR || want == MPCQP_BACKEND_BCR8 || want == MPCQP_BACKEND_BCRT;
    if (want_bcr && !bcr_shape) return refuse("BCR");
    // AUTO, measured against the bandwidth kernel (device loop and stepwise path, 1024 .. 4096 instances; LAB_NOTES.md):
    //   * horizons of 21 .. 30 steps (the 31-stage schedule: every wave owns one group of four stages and the kernel runs on five barriers per iteration, mpcqp_latw.h) with
    //     stages too wide for the bandwidth kernel to pack two per block (nx + nu > 8): at EVERY batch size -- (12,4,30) 1.52 / 1.66 / 1.72 M solves/s against 1.28 / 1.37 /
    //     1.47 M at 1024 / 2048 / 4096 instances, (12,4,25) 1.54 / 1.69 / 1.75 against 1.43 / 1.50 / 1.68, (8,8,30) 1.93 / 2.13 / 2.20 against 1.50 / 1.72 / 1.94, (6,3,28) 1.80 /
    //     1.90 / 2.02 against 1.44 / 1.77 / 1.88, (10,6,24) 1.79 / 1.88 / 1.93 against 1.60 / 1.77 / 2.01, (12,4,21) level (1.56 / 1.69 / 1.75 against 1.59 / 1.70 / 1.84);
    //   * anything else it is eligible for: up to three instances per compute unit -- at 768 instances (12,4,10) 2.32 against 2.21, (6,2,20) 1.71 / 1.61; beyond that the
    //     bandwidth kernel is ahead on short or padded schedules ((12,4,10) 2.63 against 2.36 at 1024, (12,4,14) 2.17 / 1.80) and on stages it packs ((6,2,20) 2.63 / 2.04 and
    //     (5,1,30) 1.72 / 1.63 at 2048).
    const bool bcr_any_batch = bcr_schedule(L.N) == 31 && L.N >= 22 && L.nb > 8;
    const bool bcr = bcr_shape && (want == MPCQP_BACKEND_AUTO ? (h->ncu > 0 && (bcr_any_batch || batch <= 3 * h->ncu)) : want_bcr);
    h->L.bcr = bcr ? bcr_schedule(L.N) : 0;
    // What AUTO runs it on: 512-thread workgroups (two waves per SIMD) with a dense top (mpcqp_latw.h, mpcqp_w8.hip) -- 128 / 256 / 512 instances
    // 608 k / 1.03 M / 1.13 M solves/s against 506 k / 841 k / 935 k on four waves with the plain reduction (MPCQP_BACKEND_BCR, mpcqp_lat.h).
    // MPCQP_BACKEND_BCRT: the dense-top format on four waves (533 k / 890 k: what the eight waves add on top of the format).
    const bool bcr8 = bcr && (want == MPCQP_BACKEND_BCR8 || (want == MPCQP_BACKEND_AUTO && MPCQP_AUTO_BCR8));
    const bool bcrt = bcr8 || (bcr && want == MPCQP_BACKEND_BCRT);
    h->L.bcrtop = bcrt ? BcrFmt::top_count(h->L.bcr) : 0;
    h->L.nw = bcr8 ? 8 : NWAVES;
    // Small stages (nx + nu <= 8) that neither of the register-resident backends takes: several stages per 16 x 16 block (mpcqp_group.h) -- the
    // chain and the factor shrink by the group size.  (Worth it once the chain is long: at least three super-stages.)
    const int grp = (!dense && !bcr && L.NB == 16 && group_size(L.nb) >= 2 && group_count(L.N, group_size(L.nb)) >= 3 && !(h->S.tuning & MPCQP_TUNE_NO_GROUPING)) ? group_size(L.nb) : 0;
    h->L.grp = grp;
    if (grp) {
        h->L.fstage = GroupFmt::REC; h->L.fhead = 0; h->L.ffwd = GroupFmt::NN; h->L.ftab = 0;
        h->L.tsz += (group_count(L.N, grp) + 2) * L.NB;    // the grouped vector Tg behind Tc (+ the two slots of the middle stage's inputs)
    }
    const int NS = h->L.bcr;                                        // stages of the schedule (>= L.N)
    if (bcr) h->L.tsz = std::max(h->L.tsz + NS * L.NB, LAT_LDS_DOUBLES(NS));      // the stage-major vectors of the latency round (mpcqp_lat.h); c_e of the streaming solve
    P.fsz = dense ? (long long)DenseFmt::DOUBLES : bcr ? BcrFmt::doubles(NS, bcrt) : grp ? (long long)(group_count(L.N, grp) + 1) * GroupFmt::REC
                  : (long long)L.fhead + (long long)L.N * L.fstage;
    int rc = 0;
    rc |= dalloc(h, &P.model, B * L.model_sz); rc |= dalloc(h, &P.step, B * L.step_sz);
    rc |= dalloc(h, &P.D, B * L.n); rc |= dalloc(h, &P.E, B * L.m); rc |= dalloc(h, &P.c, B);
    rc |= dalloc(h, &P.omega, B * L.m); rc |= dalloc(h, &P.s, B * L.n); rc |= dalloc(h, &P.rho, B);
    rc |= dalloc(h, &P.F, (B + 1) * (size_t)P.fsz);      // (slot B: the shared factor of mpcqp_share_factor)
    rc |= dalloc(h, &h->fown_dev, B); rc |= dalloc(h, &h->nshared_dev, 4);
    rc |= dalloc(h, &P.x, B * L.n); rc |= dalloc(h, &P.z, B * L.m); rc |= dalloc(h, &P.y, B * L.m);
    rc |= dalloc(h, &P.xo, B * L.n); rc |= dalloc(h, &P.yo, B * L.m);
    rc |= dalloc(h, &P.dx, B * L.n); rc |= dalloc(h, &P.dy, B * L.m);
    rc |= dalloc(h, &P.qv, B * (size_t)(L.n_x + L.n_u));
    if (bcr) rc |= dalloc(h, &P.bws, B * (size_t)NS * BcrFmt::WSTAGE);
    if (L.NB == 128) rc |= dalloc(h, &P.bws, B * (size_t)HugeFmt::GWS);      // (the factorization's work matrices of 128-wide stages)
    if (L.border) { rc |= dalloc(h, &P.Bb, B * (size_t)L.nu * L.N * L.NB); rc |= dalloc(h, &P.Zb, B * (size_t)L.nu * L.N * L.NB); rc |= dalloc(h, &P.Sig, B * (size_t)L.nu * L.nu); }
    rc |= dalloc(h, &P.Dt, B * L.n); rc |= dalloc(h, &P.Et, B * L.m);
    rc |= dalloc(h, &P.ctype, B * L.m); rc |= dalloc(h, &P.info, B); rc |= dalloc(h, &P.stats, 8);
        rc |= dalloc(h, &h->u0_dev, B * L.nu);
    rc |= dalloc(h, &P.work, B); rc |= dalloc(h, &h->perm_dev, B);
    rc |= dalloc(h, &h->vcur_dev, (size_t)std::max(4 * h->ncu, 1)); rc |= dalloc(h, &h->vqueue_dev, 4); rc |= dalloc(h, &h->vdone_dev, B); rc |= dalloc(h, &h->qperm_dev, B);
    rc |= dalloc(h, &P.tstamp, (size_t)TS_STRIDE * B);
    rc |= dalloc(h, &h->pending_dev, B); rc |= dalloc(h, &h->npending_dev, 4);
    if (h->S.tuning & MPCQP_TUNE_NO_BALANCE) h->auto_balance = 0;
    if (rc) { std::string msg = g_err; mpcqp_destroy(h); return fail(MPCQP_ERR_HIP, msg); }
    // The factorization's workspace starts at the work area T, the last part of the common block,
    // and may run on into the iterate area (dead while a factorization runs); T is widened only where even that is short.
    const int toplds = bcrt ? LATW_TOP_LDS(NS) + 2 : 0;      // the round's LDS copy of the top inverse, behind the iterate (+ alignment slack)
    const int fws = dense ? L.NR * L.dld + 2 * DenseFmt::ROWS + L.m + L.n : bcr ? BcrFmt::lds_doubles(h->L.nw, L.m + L.n, bcrt ? BcrFmt::top_count(NS) : 0) : (L.NB == 16 ? FactorCfg<16>::WS : L.NB == 32 ? FactorCfg<32>::WS : L.NB == 64 ? WideFmt::WS : HugeFmt::WS);
    // (a held input, Nc < Np: border_factor also forms Sigma and its inverse, 2 nu^2 doubles, at the start of T -- more than any factorization
    //  workspace once nu is large: (30, 40, 3, 2) needs 3 200 doubles where the 128-wide factorization asks for 264)
    const int need = std::max(fws, (L.border && !dense) ? 2 * L.nu * L.nu : 0);
    const int avail = L.tsz + (h->lds_state ? (int)state_doubles : 0) + toplds;
    if (avail < need) h->L.tsz += need - avail;
    h->smem_setup = sizeof(double) * ((size_t)smem_common_doubles(h->L) + (h->lds_state ? state_doubles : 0) + (size_t)toplds);      // every kernel gets the full block
    // At most one instance per compute unit and an iterate that does not qualify for the LDS-resident owner map: stage it (with the metric
    // vectors, the linear cost and the border matrices) into LDS for the length of a round if one workgroup's LDS holds it (admm_round_global).
    {
        const size_t stage_doubles = 2 * (size_t)L.n + 3 * (size_t)L.m + (size_t)(L.n_x + L.n_u) + (L.border ? 2 * (size_t)L.nu * L.N * L.NB + (size_t)L.nu * L.nu : 0);
        const bool specialised = L.NB == 32 && L.nx == 20 && L.nu == 8 && !L.border;      // (the BASELINE cfg-5 instantiation has compile-time dimensions and no staged round)
        const bool lstage = !h->lds_state && !dense && !bcr && !specialised && h->ncu > 0 && batch <= h->ncu && h->smem_setup + sizeof(double) * stage_doubles <= 150 * 1024
                            && !(h->S.tuning & MPCQP_TUNE_NO_LSTAGE);
        h->L.lstage = lstage ? 1 : 0;
        if (lstage) h->smem_setup += sizeof(double) * stage_doubles;
        // ... and of those, the ones on grouped stages (nx + nu <= 8 on a long horizon: the reference's notebook (4,1,150,75) and Kalman (4,1,200,200)
        // examples with compile-time dimensions, anything else generically) run the 512-thread instantiation of the same kernel (mpcqp_w8.hip):
        // eight waves for the owner passes that are half of their iteration
        if (lstage && h->L.grp > 1 && !(h->S.tuning & MPCQP_TUNE_NO_W8)) {
            h->smem_setup += sizeof(double) * 16 * (8 - h->L.nw);      // (the reduction scratch grows with the waves: smem_common)
            h->L.nw = 8;
        }
    }
    {   // The weight matrices [Qx | QxN | Qu | QDu] ride with the hot prefix into LDS (Lay::hot_lds) where that costs no residency: the kernels that
        // run one workgroup per compute unit (latency backends, 512 threads, wide stages) or two (32 x 32 stages), not the four-per-CU bandwidth
        // kernel of 16 x 16 stages, whose 40 KB are spoken for.
        const size_t extra = sizeof(double) * (size_t)(h->L.model_sz - h->L.hot_sz);
        const int occ = (h->L.nw == 8 || h->L.dense || h->L.bcr || h->L.NB > 32) ? 1 : h->L.NB <= 16 ? 4 : 2;
        if ((h->smem_setup + extra + 1024) * (size_t)occ <= 160 * 1024) { h->L.hot_lds = h->L.model_sz; h->smem_setup += extra; }
    }
    h->smem_solve = h->smem_setup;
    if (h->smem_solve > 160 * 1024) { mpcqp_destroy(h); return fail(MPCQP_ERR_UNSUPPORTED, "problem too large for one workgroup's LDS"); }
    // check_norms_own reads the weight matrices through LDS views (as_lds): with an LDS-resident iterate they must BE in LDS -- staged with the hot
    // prefix, or behind W in the work area.  The sizing above guarantees it; a layout change that breaks it must fail here, not give wrong residuals.
    if (h->lds_state && h->L.hot_lds == h->L.hot_sz && h->L.tsz - h->L.m < h->L.model_sz - h->L.hot_sz) {
        mpcqp_destroy(h); return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_create: internal layout error: the weight matrices of an LDS-resident handle do not fit behind W");
    }
    *out = h;
    return MPCQP_OK;
}

extern "C" void mpcqp_destroy(mpcqp_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    if (h->run_buf) hipFree(h->run_buf);
    delete h->csc;
    if (h->vec_buf) hipFree(h->vec_buf);
    if (h->pin_in) hipHostFree(h->pin_in);
    if (h->pin_out) hipHostFree(h->pin_out);
    for (int e = 0; e < h->nevents; ++e) { hipEventDestroy(h->ev0[e]); hipEventDestroy(h->ev1[e]); }
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int mpcqp_set_stream(mpcqp_handle *h, void *hip_stream) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream; h->own_stream = false;
    return MPCQP_OK;
}

extern "C" int mpcqp_synchronize(mpcqp_handle *h) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

// true if p points to device memory (results copied there are stream-ordered: the call need not wait for them)
static bool is_device_ptr(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice;
}

// Device scratch of one call: freed when the call returns, on every path.
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) hipFree(p); }
    template <class T> hipError_t get(T **out, size_t count) {
        void *q = nullptr; hipError_t e = hipMalloc(&q, count * sizeof(T)); if (e == hipSuccess) { ptrs.push_back(q); *out = (T *)q; } return e;
    }
};

// strided upload: src [batch][w] -> dst [batch][stride] at column offset off.  Host sources: one 2-D copy each.  DEVICE sources (a caller that keeps its
// data on the GPU: bench.py, the RCCL shards) are collected and moved by ONE kernel launch (flush_puts) -- a device-to-device hipMemcpy2DAsync costs
// ~ 0.2 ms of host time each, and setup() issues seventeen of them: 4 of the 5.2 ms a cold setup() of 1024 instances took.
__global__ __launch_bounds__(256) void k_pack_rows(PackArgs A) {
    for (int i = 0; i < A.n; ++i) {
        const PackItem t = A.it[i];
        const int total = A.batch * t.w;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
            const int b = idx / t.w, c = idx - b * t.w;
            t.dst[(size_t)b * t.stride + c] = t.src[idx];
        }
    }
}
static int flush_puts(mpcqp_handle *h) {
    PackArgs &A = h->pack;
    if (A.n == 0) return 0;
    A.batch = h->batch;
    int most = 1;
    for (int i = 0; i < A.n; ++i) most = std::max(most, A.it[i].w);
    const int grid = std::max(1, std::min(1024, (h->batch * most + 255) / 256));
    hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, h->stream, A);
    A.n = 0;
    HIPCHK(hipGetLastError());
    return 0;
}
static int put(mpcqp_handle *h, double *dst, int stride, int off, const double *src, int w) {
    if (is_device_ptr(src)) {
        if (h->pack.n == 24 && flush_puts(h)) return MPCQP_ERR_HIP;
        h->pack.it[h->pack.n++] = PackItem{src, dst + off, w, stride};
        return 0;
    }
    HIPCHK(hipMemcpy2DAsync(dst + off, sizeof(double) * (size_t)stride, src, sizeof(double) * (size_t)w,
                            sizeof(double) * (size_t)w, (size_t)h->batch, hipMemcpyDefault, h->stream));
    return 0;
}

#define DISPATCH_NB(NBV, EXPR) switch (NBV) { \
    case 16: { constexpr int NB = 16; EXPR; } break; \
    case 32: { constexpr int NB = 32; EXPR; } break; \
    case 64: { constexpr int NB = 64; EXPR; } break; \
    default: { constexpr int NB = 128; EXPR; } break; }

template <class K>
static int set_smem(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

static int step_upload(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows) {
    const Lay &L = h->L;
    if (x0 && put(h, h->P.step, L.step_sz, 0, x0, L.nx)) return MPCQP_ERR_HIP;
    if (um1 && put(h, h->P.step, L.step_sz, L.nx, um1, L.nu)) return MPCQP_ERR_HIP;
    if (xref) {
        if (xref_rows != 1 && xref_rows != L.N) return fail(MPCQP_ERR_ARG, "xref_rows must be 1 or Np+1");
        h->L.xref_rows = xref_rows;
        if (put(h, h->P.step, L.step_sz, L.nx + L.nu, xref, xref_rows * L.nx)) return MPCQP_ERR_HIP;
    }
    return flush_puts(h) ? MPCQP_ERR_HIP : 0;
}

static int share_factor_async(mpcqp_handle *h);
// setup's two launches (mpcqp_phases.h): equilibration, rho vector and cold start with a lean LDS block; then the first factorization
static int launch_setup(mpcqp_handle *h) {
    const Lay &L = h->L;
    h->P.fown = nullptr;                            // (every instance factors into its own slot: sharing is off until mpcqp_share_factor is asked again)
    if (flush_puts(h)) return MPCQP_ERR_HIP;
    const size_t lean = sizeof(double) * (size_t)(smem_common_doubles(L) - L.tsz);      // (the work area T is carved last and not touched by k_setup)
    const size_t de = sizeof(double) * (size_t)(L.n + L.m);
    const int lds_de = lean + de <= 96 * 1024 ? 1 : 0;      // (cfg-5: 91 KB -- one workgroup per compute unit, still well ahead of two walking memory)
    if (set_smem(k_setup, lean + (lds_de ? de : 0))) return MPCQP_ERR_HIP;
    hipLaunchKernelGGL(k_setup, dim3(h->batch), dim3(NT), lean + (lds_de ? de : 0), h->stream, h->L, h->P, h->S, lds_de);
    // the cyclic reduction's factorization at 256 threads uses the work area only as far as BcrFmt::lds_doubles says (three workgroups per compute unit at (12,4,30)
    // where the solve kernel's block -- iterate, top inverse -- would allow one); everything else factors in the block it solves in
    if (L.NB == 16 && L.bcr) {
        const size_t fac = lean + sizeof(double) * (size_t)BcrFmt::lds_doubles(NT / 64, L.m + L.n, L.bcrtop);
        if (set_smem(k_setup_factor_bcr, fac)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_setup_factor_bcr, dim3(h->batch), dim3(NT), fac, h->stream, h->L, h->P);
    } else {
        DISPATCH_NB(L.NB, {
            if (set_smem(k_setup_factor<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
            hipLaunchKernelGGL(k_setup_factor<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P);
        });
    }
    HIPCHK(hipGetLastError());
    // a batch of copies of ONE controller (the reference's one-model-many-states caller) is detected here: instances whose factorization inputs equal
    // instance 0's share one copy of its factor from the start (mpcqp_share_factor says what that means; one map kernel and a 0.5 MB copy per setup)
    if (h->batch > 1 && !(h->S.tuning & MPCQP_TUNE_NO_SHARE)) return share_factor_async(h);
    return MPCQP_OK;
}

extern "C" int mpcqp_setup(mpcqp_handle *h, const mpcqp_model *M, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h || !M || !x0 || !um1 || !xref) return fail(MPCQP_ERR_ARG, "mpcqp_setup: null argument");
    if (!M->Ad || !M->Bd || !M->Qx || !M->QxN || !M->Qu || !M->QDu || !M->xmin || !M->xmax || !M->umin || !M->umax ||
        !M->Dumin || !M->Dumax || !M->uref || !M->eps_feas) return fail(MPCQP_ERR_ARG, "mpcqp_setup: null model field");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; const int nx = L.nx, nu = L.nu, ms = L.model_sz;
    double *mb = h->P.model;
    int rc = 0;
    h->pack.n = 0;                                  // (nothing left over from a call that failed half-way)
    rc |= put(h, mb, ms, L.oAd, M->Ad, nx * nx); rc |= put(h, mb, ms, L.oBd, M->Bd, nx * nu);
    rc |= put(h, mb, ms, L.oxmin, M->xmin, nx); rc |= put(h, mb, ms, L.oxmax, M->xmax, nx);
    rc |= put(h, mb, ms, L.oumin, M->umin, nu); rc |= put(h, mb, ms, L.oumax, M->umax, nu);
    rc |= put(h, mb, ms, L.oDumin, M->Dumin, nu); rc |= put(h, mb, ms, L.oDumax, M->Dumax, nu);
    rc |= put(h, mb, ms, L.ouref, M->uref, nu); rc |= put(h, mb, ms, L.oeps, M->eps_feas, 1);
    rc |= put(h, mb, ms, L.oQx, M->Qx, nx * nx); rc |= put(h, mb, ms, L.oQxN, M->QxN, nx * nx);
    rc |= put(h, mb, ms, L.oQu, M->Qu, nu * nu); rc |= put(h, mb, ms, L.oQDu, M->QDu, nu * nu);
    if (rc) return MPCQP_ERR_HIP;
    if ((rc = step_upload(h, x0, um1, xref, xref_rows))) return rc;
    if ((rc = launch_setup(h))) return rc;
    h->is_setup = true;
    return MPCQP_OK;
}

extern "C" int mpcqp_update(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_update before mpcqp_setup");
    if (h->step_blank && (!x0 || !um1 || !xref)) return fail(MPCQP_ERR_STATE, "mpcqp_update: the handle was set up from raw q, l, u and holds no x0 / u_{-1} / xref yet: give all three");
    HIPCHK(hipSetDevice(h->device));
    h->step_blank = false;
    h->L.raw = 0;                                   // q, l, u are rebuilt from (x0, u_{-1}, xref) again
    h->pack.n = 0;
    return step_upload(h, x0, um1, xref, xref_rows);
}

// The caller's q, l, u reach the decode kernel in place if they are device memory, else through the handle's staging block
// (allocated once; copies and kernel are stream-ordered, so the next call may reuse it without waiting).
static int upload_vectors(mpcqp_handle *h, const double *q, const double *l, const double *u) {
    const Lay &L = h->L; const size_t B = (size_t)h->batch;
    const double *src[3] = {q, l, u}; const size_t cnt[3] = {B * L.n, B * L.m, B * L.m};
    const double *dev[3] = {nullptr, nullptr, nullptr};
    size_t off = 0;
    for (int i = 0; i < 3; ++i) {
        if (src[i] && !is_device_ptr(src[i])) {
            if (!h->vec_buf) HIPCHK(hipMalloc((void **)&h->vec_buf, sizeof(double) * B * (L.n + 2 * (size_t)L.m)));
            HIPCHK(hipMemcpyAsync(h->vec_buf + off, src[i], cnt[i] * sizeof(double), hipMemcpyHostToDevice, h->stream));
            dev[i] = h->vec_buf + off;
        } else dev[i] = src[i];
        off += cnt[i];
    }
    hipLaunchKernelGGL(k_decode_vectors, dim3(h->batch), dim3(64), 0, h->stream, h->L, h->P, dev[0], dev[1], dev[2], h->batch);
    HIPCHK(hipGetLastError());
    return MPCQP_OK;
}

extern "C" int mpcqp_update_vectors(mpcqp_handle *h, const double *q, const double *l, const double *u) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_update_vectors before setup");
    if ((l == nullptr) != (u == nullptr)) return fail(MPCQP_ERR_ARG, "mpcqp_update_vectors: give l and u together (the equality rows l[:nx] == u[:nx] carry x0)");
    HIPCHK(hipSetDevice(h->device));
    if (!h->L.raw) {
        // entering raw mode: the vectors that are NOT given now must keep describing the current problem, so the
        // tables behind them are materialised once from the controller data (q from (xref, uref, u_{-1}); du0 from u_{-1})
        if (!q || !l || !u) {
            DISPATCH_NB(h->L.NB, {
                if (set_smem(k_export<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
                hipLaunchKernelGGL(k_export<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
            });
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_keep_du0, dim3((h->batch * h->L.nu + 255) / 256), dim3(256), 0, h->stream, h->L, h->P, h->batch);
            HIPCHK(hipGetLastError());
        }
        h->L.raw = 1;
    }
    return upload_vectors(h, q, l, u);
}

extern "C" int mpcqp_setup_qp(mpcqp_handle *h, const mpcqp_model *M, const double *q, const double *l, const double *u) {
    if (!h || !M || !q || !l || !u) return fail(MPCQP_ERR_ARG, "mpcqp_setup_qp: null argument");
    if (!M->Ad || !M->Bd || !M->Qx || !M->QxN || !M->Qu || !M->QDu || !M->eps_feas) return fail(MPCQP_ERR_ARG, "mpcqp_setup_qp: null model field");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; const int nx = L.nx, nu = L.nu, ms = L.model_sz;
    double *mb = h->P.model;
    int rc = 0;
    h->pack.n = 0;                                  // (nothing left over from a call that failed half-way)
    rc |= put(h, mb, ms, L.oAd, M->Ad, nx * nx); rc |= put(h, mb, ms, L.oBd, M->Bd, nx * nu);
    rc |= put(h, mb, ms, L.oeps, M->eps_feas, 1);
    if (M->uref) rc |= put(h, mb, ms, L.ouref, M->uref, nu);          // (only output()'s u_failure reads it; zeros otherwise)
    rc |= put(h, mb, ms, L.oQx, M->Qx, nx * nx); rc |= put(h, mb, ms, L.oQxN, M->QxN, nx * nx);
    rc |= put(h, mb, ms, L.oQu, M->Qu, nu * nu); rc |= put(h, mb, ms, L.oQDu, M->QDu, nu * nu);
    if (rc) return MPCQP_ERR_HIP;
    h->L.raw = 1; h->step_blank = true;
    if ((rc = upload_vectors(h, q, l, u))) return rc;
    if ((rc = launch_setup(h))) return rc;
    h->is_setup = true;
    return MPCQP_OK;
}

// ---- the seam with the caller's matrices (mpc.py:266), see mpcqp_csc.h
extern "C" int mpcqp_create_csc(mpcqp_handle **out, int device, int batch, int n, int m, const int64_t *P_colptr, const int32_t *P_rowidx,
                                const int64_t *A_colptr, const int32_t *A_rowidx, int nx_hint, int nu_hint, const mpcqp_settings *s) {
    if (!out || !P_colptr || !P_rowidx || !A_colptr || !A_rowidx || n < 1 || m < 1) return fail(MPCQP_ERR_ARG, "mpcqp_create_csc: null argument");
    CscSeam *seam = new CscSeam();
    CscPattern &c = seam->pat;
    c.n = n; c.m = m;
    c.Pp.assign(P_colptr, P_colptr + n + 1); c.Ap.assign(A_colptr, A_colptr + n + 1);
    bool cp_ok = c.Pp[0] == 0 && c.Ap[0] == 0;
    for (int j = 0; j < n && cp_ok; ++j) cp_ok = c.Pp[j] <= c.Pp[j + 1] && c.Ap[j] <= c.Ap[j + 1];      // non-decreasing from 0: every entry is within [0, colptr[n]]
    if (!cp_ok) { delete seam; return fail(MPCQP_ERR_ARG, "mpcqp_create_csc: bad column pointers (must start at 0 and be non-decreasing)"); }
    c.Pi.assign(P_rowidx, P_rowidx + c.Pp[n]); c.Ai.assign(A_rowidx, A_rowidx + c.Ap[n]);
    for (int32_t r : c.Pi) if (r < 0 || r >= n) { delete seam; return fail(MPCQP_ERR_ARG, "mpcqp_create_csc: P row index out of range"); }
    for (int32_t r : c.Ai) if (r < 0 || r >= m) { delete seam; return fail(MPCQP_ERR_ARG, "mpcqp_create_csc: A row index out of range"); }
    int nx, nu, Np, Nc, soft; std::string why;
    if (csc_dims(c, nx_hint, nu_hint, &nx, &nu, &Np, &Nc, &soft, &why)) { delete seam; return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_create_csc: not an MPC QP of pyMPC: " + why); }
    mpcqp_settings st; if (s) st = *s; else mpcqp_default_settings(&st);
    st.soft_constraints = soft;                     // (slack columns or not: read off the pattern by csc_dims)
    const int rc = mpcqp_create(out, device, batch, nx, nu, Np, Nc, &st);
    if (rc) { delete seam; return rc; }
    (*out)->csc = seam;
    return MPCQP_OK;
}

extern "C" int mpcqp_setup_csc(mpcqp_handle *h, const double *P_val, const double *A_val, const double *q, const double *l, const double *u) {
    if (!h || !P_val || !A_val || !q || !l || !u) return fail(MPCQP_ERR_ARG, "mpcqp_setup_csc: null argument");
    if (!h->csc) return fail(MPCQP_ERR_STATE, "mpcqp_setup_csc: the handle was not made by mpcqp_create_csc");
    const CscPattern &c = h->csc->pat; const Lay &L = h->L;
    const size_t B = (size_t)h->batch, nnzP = (size_t)c.Pp[c.n], nnzA = (size_t)c.Ap[c.n];
    const int nx = L.nx, nu = L.nu;
    std::vector<double> Ad(B * nx * nx), Bd(B * nx * nu), Qx(B * nx * nx), QxN(B * nx * nx), Qu(B * nu * nu), QDu(B * nu * nu), ef(B), lc(B * c.m), uc(B * c.m);
    for (size_t b = 0; b < B; ++b) {
        MpcBlocks blk; std::string why;
        if (csc_recover(c, nx, nu, L.Np, L.Nc, L.soft, P_val + b * nnzP, A_val + b * nnzA, q + b * c.n, l + b * c.m, u + b * c.m, &blk, &why))
            return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_setup_csc: instance " + std::to_string(b) + ": " + why);
        std::copy(blk.Ad.begin(), blk.Ad.end(), Ad.begin() + b * nx * nx); std::copy(blk.Bd.begin(), blk.Bd.end(), Bd.begin() + b * nx * nu);
        std::copy(blk.Qx.begin(), blk.Qx.end(), Qx.begin() + b * nx * nx); std::copy(blk.QxN.begin(), blk.QxN.end(), QxN.begin() + b * nx * nx);
        std::copy(blk.Qu.begin(), blk.Qu.end(), Qu.begin() + b * nu * nu); std::copy(blk.QDu.begin(), blk.QDu.end(), QDu.begin() + b * nu * nu);
        ef[b] = blk.eps_feas;
    }
    for (size_t i = 0; i < B * c.m; ++i) { lc[i] = std::min(std::max(l[i], -QP_INFTY), QP_INFTY); uc[i] = std::min(std::max(u[i], -QP_INFTY), QP_INFTY); }      // (osqp's wrapper clips to +-1e30)
    mpcqp_model M; memset(&M, 0, sizeof(M));
    M.Ad = Ad.data(); M.Bd = Bd.data(); M.Qx = Qx.data(); M.QxN = QxN.data(); M.Qu = Qu.data(); M.QDu = QDu.data(); M.eps_feas = ef.data();
    const int rc = mpcqp_setup_qp(h, &M, q, lc.data(), uc.data());
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));        // (the host staging vectors above die with this call)
    return MPCQP_OK;
}

extern "C" int mpcqp_update_settings(mpcqp_handle *h, const mpcqp_settings *s) {
    if (!h || !s) return fail(MPCQP_ERR_ARG, "null argument");
    double rho = h->S.rho, sigma = h->S.sigma; int scaling = h->S.scaling, backend = h->S.backend, tuning = h->S.tuning;
    h->S = *s;
    h->S.rho = rho; h->S.sigma = sigma; h->S.scaling = scaling;   // fixed at setup (they shape the factorization)
    h->S.soft_constraints = h->L.soft;                            // fixed at creation (it shapes the problem)
    h->S.backend = backend; h->S.tuning = tuning;                 // ... as are the backend and its tuning flags
    return MPCQP_OK;
}

// The kernel arguments of a launch and its grid: one workgroup per instance, or -- batches beyond the resident workgroup slots -- the PERSISTENT form:
// as many workgroups as there are slots, each taking instances off a queue (k_mpc_run in mpcqp_kernels.h says what that buys).
// nsteps: closed-loop steps of the launch (0: a solve).
// Fewer slots than fit for the bandwidth kernel of 32 x 32 stages with its iterate in memory (BASELINE configs[4]) where a closed-loop launch would otherwise hold
// (nearly) every instance resident at once: such a launch ends with its slowest instance, which -- sharing the memory system with 511 others -- runs at 104 us per
// iteration where it could run at 75.  Three quarters of the slots and a queue of (instance, step range) items, longest expected work first: the stragglers start
// first and run faster all along, the memory system stays saturated.  Measured (50-step launches, k solves/s, 4 / 5 / 6 / 7 / 8 quarters of a workgroup per unit):
// 512 instances 121 / 138 / 148-150 / 141-144 / 137-138; 768 instances - / - / 153 / - / 145; 1024 instances 119 / - / 153 / 166 / 177 (two full rounds: all slots).
static int run_grid(const mpcqp_handle *h, int nsteps) {
    const Lay &L = h->L;
    const bool one_at_a_time = L.bcr || L.dense || L.NB > 32 || L.nw == 8;
    int occ = one_at_a_time ? 1 : (L.NB <= 16 ? 4 : 2);
    if (h->smem_solve > 0) occ = std::max(1, std::min(occ, (int)((size_t)160 * 1024 / h->smem_solve)));
    const int full = h->ncu * occ;
    const int q = (h->S.tuning >> MPCQP_TUNE_SLOTS_SHIFT) & 0x1F;      // (development: resident workgroups in eighths of a workgroup per compute unit)
    if (q) return std::max(1, std::min(full, h->ncu * q / 8));
    if (L.NB == 32 && occ == 2 && !h->lds_state && !L.lstage && nsteps >= 8 && 2 * h->batch <= 3 * full && 4 * h->batch > 3 * full && !(h->S.tuning & MPCQP_TUNE_NO_QUEUE)) return 3 * full / 4;
    return full;
}
static RunKArgs run_kernel_args(mpcqp_handle *h, const RunArgs &R0, int *grid) {
    RunKArgs A; A.L = h->L; A.P = h->P; A.S = h->S; A.R = R0;
    *grid = h->batch;
    const int slots = run_grid(h, R0.nsteps);
    if (h->ncu > 0 && h->batch > slots && h->vcur_dev && !R0.pin_in && !R0.pub && (R0.nsteps > 0 || (h->S.tuning & (MPCQP_TUNE_QUEUE_SOLVES | MPCQP_TUNE_ONE_LAUNCH_SOLVES))) && !(h->S.tuning & MPCQP_TUNE_NO_QUEUE)) {
        A.R.vcur = h->vcur_dev; A.R.vperm = ((R0.nsteps > 0 || (R0.part == 0 && (h->S.tuning & MPCQP_TUNE_ONE_LAUNCH_SOLVES))) && h->qperm_set) ? h->qperm_dev : h->P.perm;      /* (a solve's second launch walks the pending list: P.perm) */ A.R.vqueue = h->vqueue_dev;
        A.P.perm = h->vcur_dev;
        *grid = slots;
        // closed loop: an instance's steps in parts, about QUEUE_ITEMS_PER_SLOT items per slot in all (a launch ends within half an item of its ideal
        // length; an item costs a kernel prologue, ~ 10 us)
        if (R0.nsteps > 1 && !(h->S.tuning & MPCQP_TUNE_NO_PARTS)) {
            const int per_slot = ((h->S.tuning >> 16) & 0xFF) ? ((h->S.tuning >> 16) & 0xFF) : QUEUE_ITEMS_PER_SLOT;      // (bits 16..23: development override)
            const int parts = std::min(R0.nsteps, (per_slot * slots + h->batch - 1) / h->batch);
            if (parts > 1) {
                // part lengths fall off towards the end of the loop (each takes about 2 / (parts left + 1) of what is left: 20 steps in 4 parts = 8, 6, 4, 2): the
                // launch's end is as fine-grained as many small parts would make it, for the prologues of few
                const int np = std::min(parts, 16);
                int off = 0;
                for (int p = 0; p < np; ++p) {
                    A.R.voff[p] = off;
                    const int left = R0.nsteps - off, pl = np - p;
                    int len = pl == 1 ? left : std::max(1, std::min(left - (pl - 1), (2 * left + pl) / (pl + 1)));
                    if (h->S.tuning & MPCQP_TUNE_EVEN_PARTS) len = pl == 1 ? left : (left + pl - 1) / pl;
                    off += len;
                }
                A.R.voff[np] = R0.nsteps; A.R.vparts = np; A.R.vdone = h->vdone_dev;
            }
        }
    }
    return A;
}
template <int NB, bool LDSS, int NXT, int NUT, int MODE>
static int launch_run_t(mpcqp_handle *h, const RunArgs &R) {
    int grid;
    const RunKArgs A = run_kernel_args(h, R, &grid);
    if (A.R.vdone && hipMemsetAsync(h->vdone_dev, 0, sizeof(int) * (size_t)h->batch, h->stream) != hipSuccess) return MPCQP_ERR_HIP;
    if (R.nsteps > 0) {
        if (set_smem(k_mpc_run<NB, LDSS, NXT, NUT, MODE, true>, h->smem_solve)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL((k_mpc_run<NB, LDSS, NXT, NUT, MODE, true>), dim3(grid), dim3(NT), h->smem_solve, h->stream, A);
    } else {
        if (set_smem(k_mpc_run<NB, LDSS, NXT, NUT, MODE, false>, h->smem_solve)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL((k_mpc_run<NB, LDSS, NXT, NUT, MODE, false>), dim3(grid), dim3(NT), h->smem_solve, h->stream, A);
    }
    return 0;
}
template <int NB, bool LDSS>
static int launch_run_generic(mpcqp_handle *h, const RunArgs &R) {
    return h->L.border ? launch_run_t<NB, LDSS, 0, 0, MODE_BORDER>(h, R) : launch_run_t<NB, LDSS, 0, 0, MODE_CHAIN>(h, R);
}

// Load balancing across CUs.  All workgroups of a launch are resident at once (a few per CU) and an instance keeps its
// character -- one that needs two ADMM rounds per solve mostly keeps needing them -- so CUs that happen to host several
// slow instances finish last and set the launch time.  Every now and then the per-instance iteration counts are read
// back, smoothed, and the workgroup -> instance map is rebuilt: instances sorted by expected work, dealt to the CUs in
// snake order (blocks b, b + #CU, b + 2 #CU, ... share a CU: dispatch is round-robin over XCDs and CUs), so that every CU
// gets a similar total.  Pure scheduling: which workgroup handles which instance never changes a result.
// Must be called with the stream idle.
static bool balance_due(const mpcqp_handle *h) {
    if (!h->auto_balance || h->ncu <= 0 || h->batch <= h->ncu) return false;      // (nothing to balance: no call waits for the stream on this account)
    return h->solves_since_balance >= (h->work_ema.empty() ? BALANCE_FIRST : BALANCE_EVERY);
}
static int rebalance(mpcqp_handle *h) {
    const int B = h->batch, ncu = h->ncu;
    h->solves_since_balance = 0;
    if (!h->auto_balance || ncu <= 0 || B <= ncu) return MPCQP_OK;
    std::vector<unsigned> work(B);
    HIPCHK(hipMemcpyAsync(work.data(), h->P.work, sizeof(unsigned) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemsetAsync(h->P.work, 0, sizeof(unsigned) * (size_t)B, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));       // (the handle's own stream only: no implicit device-wide sync)
    if (h->work_ema.empty()) h->work_ema.assign(B, 0.0);
    bool any = false;
    const double decay = 0.75;                             // (anything from 0.5 to 0.95 measured the same)
    for (int i = 0; i < B; ++i) { h->work_ema[i] = decay * h->work_ema[i] + (double)work[i]; any |= work[i] != 0; }
    if (!any) return MPCQP_OK;
    std::vector<int> order(B), perm(B);
    for (int i = 0; i < B; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h->work_ema[a] > h->work_ema[b]; });
    // One workgroup per compute unit at a time (the latency kernels, wide stages): workgroups start in index order as compute units come
    // free, so the map is the longest-expected-work-first list -- the classic greedy schedule, makespan within one instance of the mean.
    const bool one_at_a_time = h->L.bcr || h->L.dense || h->L.NB > 32 || h->L.nw == 8;
    const int full_rows = B / ncu;
    for (int j = 0; j < B; ++j) {
        const int row = j / ncu, pos = j % ncu;
        const bool reversed = !one_at_a_time && (row & 1) && row < full_rows;      // a partial last row keeps forward order
        perm[row * ncu + (reversed ? ncu - 1 - pos : pos)] = order[j];
    }
    // (the closed loop's persistent launches take instances off a QUEUE: for them the longest-expected-work-first list itself, whatever the kernel)
    HIPCHK(hipMemcpyAsync(h->qperm_dev, order.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    h->qperm_set = true;
    // Pacing (development switch, off unless mpcqp_settings.tuning bits 8..15 ask for it): in the bandwidth kernels with a global-memory iterate whose
    // instances are ALL resident at once (B <= slots) the launch ends with its slowest instance.  Every instance that is NOT expected to need more than
    // 1.15 x the median's iterations idles `units` x 3.5 us in each of its iterations, leaving its share of the memory system to the stragglers.
    {
        const int occ = one_at_a_time ? 1 : (h->L.NB <= 16 ? 4 : 2);
        const int units = std::min((h->S.tuning >> 8) & 0xFF, 64);
        if (units && !one_at_a_time && !h->lds_state && !h->L.lstage && B <= occ * ncu) {
            const double wmed = h->work_ema[order[B / 2]];
            for (int j = 0; j < B; ++j) {
                const int inst = perm[j] & PERM_INST_MASK;
                perm[j] = inst | ((h->work_ema[inst] > 1.15 * wmed ? 0 : units) << PERM_PACE_SHIFT);
            }
        }
    }
    HIPCHK(hipMemcpyAsync(h->perm_dev, perm.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));                          // (perm is a stack-lifetime host buffer)
    h->P.perm = h->perm_dev;
    return MPCQP_OK;
}

// One launch of the solve / closed-loop kernel on the handle's stream (asynchronous).  Specialisations with
// compile-time nx, nu for the BASELINE configurations; generic kernels otherwise.
static int launch_run(mpcqp_handle *h, RunArgs R, int plain_iters) {
    const Lay &L = h->L; const mpcqp_settings &S = h->S;
    R.plain = plain_iters > 0;
    R.warm_x = (h->warm_x_pending && R.part != 2 && R.part != 3) ? 1 : 0;
    if (R.part == 0 || R.part == 2) h->warm_x_pending = false;      // (a two-launch solve begins in its first launch only; mpcqp_refactor, part 3, is not a solve)
    R.max_iter = R.plain ? plain_iters : S.max_iter;
    R.chk = R.plain ? 0 : S.check_termination;
    R.rho_every = (!R.plain && S.adaptive_rho) ? (S.adaptive_rho_interval ? S.adaptive_rho_interval : (R.chk ? 4 * R.chk : 100)) : 0;
    R.batch = h->batch;
    h->solves_since_balance = std::min(h->solves_since_balance + (R.nsteps > 0 ? R.nsteps : 1), 1 << 20);
    const int e = h->ev_count % MAXEV;
    if (h->profiling) {
        if (h->ev_count >= MAXEV) {                 // ring full: bank the oldest pair first
            float ms = 0.f; HIPCHK(hipEventSynchronize(h->ev1[e])); HIPCHK(hipEventElapsedTime(&ms, h->ev0[e], h->ev1[e]));
            h->run_ms += ms; h->run_launches += 1;
        }
        HIPCHK(hipEventRecord(h->ev0[e], h->stream));
    }
    int rc;
    if (L.dense) rc = launch_run_t<16, true, 0, 0, MODE_DENSE>(h, R);
    else if (L.nw == 8) {                           // 512-thread workgroups: the other translation unit (cyclic reduction with a dense top, or one long-horizon controller on grouped stages)
        int grid;
        const RunKArgs A = run_kernel_args(h, R, &grid);
        if (A.R.vdone) HIPCHK(hipMemsetAsync(h->vdone_dev, 0, sizeof(int) * (size_t)h->batch, h->stream));
        rc = L.bcr ? mpcqp_w8_launch(&A, sizeof(A), L.bcr == 31 && L.nx == 12 && L.nu == 4, L.bcr, R.nsteps > 0, grid, h->smem_solve, h->stream)
                   : mpcqp_w8_launch(&A, sizeof(A), L.border, 0, R.nsteps > 0, grid, h->smem_solve, h->stream);
        if (rc) return fail(MPCQP_ERR_HIP, "mpcqp_w8_launch failed");
    }
    else if (L.bcrtop && L.bcr == 31 && L.nx == 12 && L.nu == 4) rc = launch_run_t<16, true, 12, 4, MODE_BCRT + 31>(h, R);
    else if (L.bcrtop && L.bcr == 31) rc = launch_run_t<16, true, 0, 0, MODE_BCRT + 31>(h, R);
    else if (L.bcrtop && L.bcr == 21) rc = launch_run_t<16, true, 0, 0, MODE_BCRT + 21>(h, R);
    else if (L.bcrtop && L.bcr == 11) rc = launch_run_t<16, true, 0, 0, MODE_BCRT + 11>(h, R);
    else if (L.bcr == 31 && L.nx == 12 && L.nu == 4) rc = launch_run_t<16, true, 12, 4, MODE_BCR + 31>(h, R);
    else if (L.bcr == 31) rc = launch_run_t<16, true, 0, 0, MODE_BCR + 31>(h, R);
    else if (L.bcr == 21) rc = launch_run_t<16, true, 0, 0, MODE_BCR + 21>(h, R);
    else if (L.bcr == 11) rc = launch_run_t<16, true, 0, 0, MODE_BCR + 11>(h, R);
    else if (L.NB == 16 && h->lds_state && L.nx == 12 && L.nu == 4 && !L.border) rc = launch_run_t<16, true, 12, 4, MODE_CHAIN>(h, R);
    else if (L.NB == 32 && !h->lds_state && L.nx == 20 && L.nu == 8 && !L.border) rc = launch_run_t<32, false, 20, 8, MODE_CHAIN>(h, R);
    // (the reference's cart pole -- its example system, and the Kalman notebook's long horizon -- with compile-time dimensions in the global-iterate kernel)
    else if (L.NB == 16 && !h->lds_state && L.nx == 4 && L.nu == 1) rc = L.border ? launch_run_t<16, false, 4, 1, MODE_BORDER>(h, R) : launch_run_t<16, false, 4, 1, MODE_CHAIN>(h, R);
    else if (L.NB == 16) rc = h->lds_state ? launch_run_generic<16, true>(h, R) : launch_run_generic<16, false>(h, R);
    else if (L.NB == 32) rc = h->lds_state ? launch_run_generic<32, true>(h, R) : launch_run_generic<32, false>(h, R);
    else if (L.NB == 64) rc = launch_run_generic<64, false>(h, R);
    else rc = launch_run_generic<128, false>(h, R);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    if (h->profiling) { HIPCHK(hipEventRecord(h->ev1[e], h->stream)); h->ev_count += 1; }
    return MPCQP_OK;
}

// Collect the pending event pairs (synchronises on them).
static int drain_events(mpcqp_handle *h) {
    const int pending = h->ev_count < MAXEV ? h->ev_count : MAXEV;
    for (int i = 0; i < pending; ++i) {
        const int e = (h->ev_count - pending + i) % MAXEV;
        float ms = 0.f; HIPCHK(hipEventSynchronize(h->ev1[e])); HIPCHK(hipEventElapsedTime(&ms, h->ev0[e], h->ev1[e]));
        h->run_ms += ms; h->run_launches += 1;
    }
    h->ev_count = 0;
    return MPCQP_OK;
}

// One solve of every instance.  Batches larger than the number of CUs are solved in two launches: begin + first ADMM
// round + check for everybody, then the instances that are not finished (typically 40 %) are continued by a second launch
// whose workgroup -> instance map is the list the first one compiled -- the dispatcher spreads them evenly over the CUs,
// whereas staying put would leave some CUs with four running workgroups and others with none (the launch would last as
// long as two fully contended rounds).  Both launches are asynchronous; no host round trip in between.
static int launch_solve(mpcqp_handle *h, int plain_iters) {
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "solve before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    RunArgs R; memset(&R, 0, sizeof(R));
    const bool split = plain_iters == 0 && h->auto_balance && h->ncu > 0 && h->batch > h->ncu && !(h->S.tuning & MPCQP_TUNE_ONE_LAUNCH_SOLVES);
    if (!split) return launch_run(h, R, plain_iters);
    R.pending = h->pending_dev; R.npending = h->npending_dev;
    HIPCHK(hipMemsetAsync(h->npending_dev, 0, sizeof(int), h->stream));
    R.part = 1;
    int rc = launch_run(h, R, 0);
    if (rc) return rc;
    R.part = 2;
    const int *perm = h->P.perm;
    h->P.perm = h->pending_dev;                     // the follow-up launch walks the pending list
    h->solves_since_balance -= 1;                   // (one solve, two launches)
    rc = launch_run(h, R, 0);
    h->P.perm = perm;
    return rc;
}

extern "C" int mpcqp_solve(mpcqp_handle *h) { if (!h) return fail(MPCQP_ERR_ARG, "null handle"); return launch_solve(h, 0); }
extern "C" int mpcqp_refactor(mpcqp_handle *h) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_refactor before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    RunArgs R; memset(&R, 0, sizeof(R));
    R.part = 3;
    const int since = h->solves_since_balance;
    const bool prof = h->profiling;
    h->profiling = false;                           // (not a solve: keep it out of the k_mpc_run timing and the balancing clock)
    const int rc = launch_run(h, R, 0);
    h->profiling = prof; h->solves_since_balance = since;
    return rc;
}
// One model, many states (test_scripts/example_mpc_function.py:105-111): instances whose factorization inputs are bit-identical to instance 0's solve with ONE
// copy of its factor -- a streamed factor then comes out of L2 instead of HBM.  k_share_map decides per instance; results cannot change: the factorization is a
// deterministic function of (model, rho vector, scaling, cost scale), and an instance that refactors (rho update, changed constraint types) leaves the shared
// slot for its own before it writes.
__global__ __launch_bounds__(NT) void k_share_map(Lay L, Ptrs P, int *fown, unsigned *nshared, int batch) {
    const int b = blockIdx.x, tid = threadIdx.x;
    typedef const unsigned long long cu64;
    int diff = 0;
    auto cmp = [&](const double *base, size_t len) {
        cu64 *a = (cu64 *)(base + (size_t)b * len), *r = (cu64 *)base;
        for (size_t i = tid; i < len; i += NT) diff |= a[i] != r[i];
    };
    cmp(P.model, L.model_sz); cmp(P.omega, L.m); cmp(P.s, L.n); cmp(P.c, 1);
    if (P.info[b].status == MPCQP_NON_CVX && P.info[b].iter == 0) diff = 1;      // (a factorization k_setup reported bad stays where it is)
    diff = __syncthreads_or(diff);
    if (tid == 0) { fown[b] = diff ? b : batch; if (!diff) atomicAdd(nshared, 1u); }
}
static int share_factor_async(mpcqp_handle *h) {
    h->P.fown = nullptr;
    // the register-resident backends read their factor once per launch (nothing to gain); the shared slot holds instance 0's factor as of NOW
    if (h->L.dense || h->L.bcr) return MPCQP_OK;
    if (flush_puts(h)) return MPCQP_ERR_HIP;
    HIPCHK(hipMemcpyAsync(h->P.F + (size_t)h->batch * h->P.fsz, h->P.F, sizeof(double) * (size_t)h->P.fsz, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->nshared_dev, 0, sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(k_share_map, dim3(h->batch), dim3(NT), 0, h->stream, h->L, h->P, h->fown_dev, h->nshared_dev, h->batch);
    HIPCHK(hipGetLastError());
    h->P.fown = h->fown_dev;
    return MPCQP_OK;
}
extern "C" int mpcqp_share_factor(mpcqp_handle *h, int *nshared) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_share_factor before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    if (nshared) *nshared = 0;
    const int rc = share_factor_async(h);
    if (rc) return rc;
    if (nshared && h->P.fown) {
        unsigned n = 0;
        HIPCHK(hipMemcpyAsync(&n, h->nshared_dev, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        *nshared = (int)n;
    }
    return MPCQP_OK;
}
extern "C" int mpcqp_iterate(mpcqp_handle *h, int iters) {
    if (!h || iters < 1) return fail(MPCQP_ERR_ARG, "mpcqp_iterate: bad argument");
    return launch_solve(h, iters);
}

static int get(mpcqp_handle *h, void *dst, const void *src, size_t bytes) {
    if (!dst) return 0;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    return 0;
}

extern "C" int mpcqp_mpc_loop(mpcqp_handle *h, int nsteps, const mpcqp_loop *io) {
    if (!h || !io || nsteps < 1) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: bad argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_loop before mpcqp_setup");
    if ((io->Ap == nullptr) != (io->Bp == nullptr)) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: give both Ap and Bp or neither");
    const int ny = io->ny;
    if (ny < 0 || ny > 64) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: ny must be in 0..64");
    if (ny && (!io->C || !io->Lgain || !io->x_true)) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: output feedback needs C, Lgain and x_true");
    if (io->xref_traj && io->xref_rows != 0 && io->xref_rows != 1 && io->xref_rows != h->L.N)
        return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: xref_rows must be 0 (as last uploaded), 1 or Np+1");
    HIPCHK(hipSetDevice(h->device));
    if (h->L.raw) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_loop: the handle holds raw q, l, u (mpcqp_update_vectors); call mpcqp_update first");
    if (io->xref_traj && io->xref_rows) h->L.xref_rows = io->xref_rows;     // update(x, u, xref_k) with this reference shape
    const Lay &L = h->L;
    const size_t B = (size_t)h->batch, K = (size_t)nsteps, nx = L.nx, nu = L.nu;
    const size_t xblk = (size_t)L.xref_rows * nx;
    // Buffers the caller keeps in DEVICE memory are used in place (the kernel reads and writes them directly, the call is
    // stream-ordered and returns without waiting); host buffers go through one staging block on the device: inputs are
    // copied in before the launch, outputs copied out after it, and the call returns when they have arrived.
    struct Part { const void *src; void *dst; size_t bytes; size_t off; bool direct; };
    Part parts[16]; int np = 0; size_t total = 0; bool any_host = false;
    auto add = [&](const void *src, void *dst, size_t bytes) {
        const void *user = src ? src : dst;
        const bool direct = bytes && user && is_device_ptr(user);
        parts[np] = Part{src, dst, bytes, total, direct};
        if (bytes && !direct) { total += (bytes + 15) & ~size_t(15); any_host = true; }      // (a host INPUT must stay valid until its staged copy has been made: the call waits for that too)
        return np++;
    };
    const int iw = add(io->w, nullptr, io->w ? 8 * K * B * nx : 0);
    const int iA = add(io->Ap, nullptr, io->Ap ? 8 * B * nx * nx : 0), iB = add(io->Bp, nullptr, io->Bp ? 8 * B * nx * nu : 0);
    const int ir = add(io->xref_traj, nullptr, io->xref_traj ? 8 * K * B * xblk : 0);
    const int iC = add(io->C, nullptr, ny ? 8 * B * ny * nx : 0), iL = add(io->Lgain, nullptr, ny ? 8 * B * nx * ny : 0);
    const int iv = add(io->v, nullptr, (ny && io->v) ? 8 * K * B * ny : 0);
    const int ixt = add(io->x_true, io->x_true, ny ? 8 * B * nx : 0);
    const int ox = add(nullptr, io->x_traj, 8 * (K + 1) * B * nx), oxh = add(nullptr, io->xhat_traj, ny ? 8 * (K + 1) * B * nx : 0);
    const int oy = add(nullptr, io->y_traj, ny ? 8 * K * B * ny : 0), ou = add(nullptr, io->u_traj, 8 * K * B * nu);
    const int os = add(nullptr, io->status_traj, 4 * K * B), oi = add(nullptr, io->iter_traj, 4 * K * B);
    if (total > h->run_bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->run_buf) hipFree(h->run_buf);
        h->run_buf = nullptr; h->run_bytes = 0;
        HIPCHK(hipMalloc(&h->run_buf, total));
        h->run_bytes = total;
    }
    char *base = (char *)h->run_buf;
    auto dev = [&](int i) -> void * {
        if (!parts[i].bytes) return nullptr;
        if (parts[i].direct) return parts[i].src ? const_cast<void *>(parts[i].src) : parts[i].dst;
        return base + parts[i].off;
    };
    for (int i = 0; i < np; ++i)
        if (parts[i].src && parts[i].bytes && !parts[i].direct) HIPCHK(hipMemcpyAsync(dev(i), parts[i].src, parts[i].bytes, hipMemcpyDefault, h->stream));
    RunArgs R; memset(&R, 0, sizeof(R));
    R.nsteps = nsteps;
    R.w = (const double *)dev(iw); R.Ap = (const double *)dev(iA); R.Bp = (const double *)dev(iB);
    R.xref_traj = (const double *)dev(ir); R.xref_blk = (int)xblk;
    R.ny = ny; R.C = (const double *)dev(iC); R.Lg = (const double *)dev(iL); R.v = (const double *)dev(iv); R.x_true = (double *)dev(ixt);
    R.x_traj = (double *)dev(ox); R.xhat_traj = (double *)dev(oxh); R.y_traj = (double *)dev(oy); R.u_traj = (double *)dev(ou);
    R.status_traj = (int *)dev(os); R.iter_traj = (int *)dev(oi);
    int rc = launch_run(h, R, 0);
    if (rc) return rc;
    for (int i = 0; i < np; ++i)
        if (parts[i].dst && parts[i].bytes && !parts[i].direct && get(h, parts[i].dst, dev(i), parts[i].bytes)) return MPCQP_ERR_HIP;
    const bool due = balance_due(h);
    if (!due && !any_host) return MPCQP_OK;                   // every buffer is device memory: stream-ordered, nothing to wait for
    HIPCHK(hipStreamSynchronize(h->stream));
    return due ? rebalance(h) : MPCQP_OK;
}

extern "C" int mpcqp_mpc_run(mpcqp_handle *h, int nsteps, const double *w, const double *Ap, const double *Bp,
                             double *x_traj, double *u_traj, int32_t *status_traj, int32_t *iter_traj) {
    mpcqp_loop io; memset(&io, 0, sizeof(io));
    io.w = w; io.Ap = Ap; io.Bp = Bp; io.x_traj = x_traj; io.u_traj = u_traj; io.status_traj = status_traj; io.iter_traj = iter_traj;
    return mpcqp_mpc_loop(h, nsteps, &io);
}

// The equality-constrained QP of the handle's problem (dynamics rows only) by at most `sweeps` multiplier sweeps in residual form, see k_eq_solve.
extern "C" int mpcqp_eq_solve(mpcqp_handle *h, int sweeps, int cold, double tol, double *res) {
    if (!h || sweeps < 0) return fail(MPCQP_ERR_ARG, "mpcqp_eq_solve: bad argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_eq_solve before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    double *dres = nullptr;
    Scratch sc;
    HIPCHK(sc.get(&dres, (size_t)h->batch * 5));
    DISPATCH_NB(h->L.NB, {
        if (set_smem(k_eq_solve<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_eq_solve<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, sweeps, cold, tol, dres);
    });
    HIPCHK(hipGetLastError());
    if (res) HIPCHK(hipMemcpyAsync(res, dres, sizeof(double) * 5 * (size_t)h->batch, hipMemcpyDefault, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_get_solution(mpcqp_handle *h, double *x, double *y, mpcqp_info *info) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    size_t B = (size_t)h->batch;
    if (get(h, x, h->P.xo, B * h->L.n * sizeof(double)) || get(h, y, h->P.yo, B * h->L.m * sizeof(double)) ||
        get(h, info, h->P.info, B * sizeof(mpcqp_info))) return MPCQP_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_get_u0(mpcqp_handle *h, double *u0) {
    if (!h || !u0) return fail(MPCQP_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    int tot = h->batch * h->L.nu;
    hipLaunchKernelGGL(k_gather_u0, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->L, h->P.xo, h->u0_dev, h->batch);
    HIPCHK(hipGetLastError());
    if (get(h, u0, h->u0_dev, sizeof(double) * (size_t)tot)) return MPCQP_ERR_HIP;
    const bool due = balance_due(h);
    if (!due && is_device_ptr(u0)) return MPCQP_OK;           // device destination: stream-ordered, no need to wait
    HIPCHK(hipStreamSynchronize(h->stream));
    return due ? rebalance(h) : MPCQP_OK;
}

extern "C" int mpcqp_mpc_step(mpcqp_handle *h, const double *x0, const double *uminus1, const double *xref, int xref_rows, double *u_out) {
    if (!h || !x0 || !u_out) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_step: null argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_step before mpcqp_setup");
    if (h->step_blank && (!uminus1 || !xref)) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_step: the handle was set up from raw q, l, u and holds no u_{-1} / xref yet: give them");
    HIPCHK(hipSetDevice(h->device));
    h->step_blank = false;
    h->L.raw = 0;
    int rc = step_upload(h, x0, uminus1, xref, xref_rows);
    if (rc) return rc;
    if ((rc = launch_solve(h, 0))) return rc;
    const int tot = h->batch * h->L.nu;
    hipLaunchKernelGGL(k_output_u, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->L, h->P, h->u0_dev, h->batch, 1);
    HIPCHK(hipGetLastError());
    if (get(h, u_out, h->u0_dev, sizeof(double) * (size_t)tot)) return MPCQP_ERR_HIP;
    const bool due = balance_due(h);
    if (!due && is_device_ptr(u_out)) return MPCQP_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    return due ? rebalance(h) : MPCQP_OK;
}

// One control step with HOST data and the lowest latency the library can offer (the drop-in class's update(): mpc.py:338-375
// = prob.update(l, u, q) + prob.solve() + reading res.x): x0 / u_{-1} / xref are placed in mapped host memory the kernel reads
// itself, the solution comes back the same way, and the call waits on a flag in that memory -- ONE kernel launch, no copy calls, no
// stream synchronisation.  Falls back to mpcqp_update + mpcqp_solve + mpcqp_get_solution where the batch is solved in two launches.
static int ensure_pinned(mpcqp_handle *h) {
    if (h->pin_tried) return h->pin_in && h->pin_out ? 0 : 1;
    h->pin_tried = true;
    const Lay &L = h->L; const size_t B = (size_t)h->batch;
    h->pin_stride = L.nx + L.nu + L.N * L.nx;
    const size_t in_bytes = sizeof(double) * B * h->pin_stride, out_bytes = sizeof(double) * B * (L.n + L.m) + sizeof(mpcqp_info) * B + 64;
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
    if (hipHostMalloc((void **)&h->pin_in, in_bytes, flags) != hipSuccess) { h->pin_in = nullptr; (void)hipGetLastError(); return 1; }
    if (hipHostMalloc((void **)&h->pin_out, out_bytes, flags) != hipSuccess) { hipHostFree(h->pin_in); h->pin_in = h->pin_out = nullptr; (void)hipGetLastError(); return 1; }
    memset(h->pin_out, 0, out_bytes);
    if (hipHostGetDevicePointer(&h->pin_in_dev, h->pin_in, 0) != hipSuccess || hipHostGetDevicePointer(&h->pin_out_dev, h->pin_out, 0) != hipSuccess) {
        hipHostFree(h->pin_in); hipHostFree(h->pin_out); h->pin_in = h->pin_out = nullptr; (void)hipGetLastError(); return 1;
    }
    return 0;
}

extern "C" int mpcqp_step_host(mpcqp_handle *h, const double *x0, const double *uminus1, const double *xref, int xref_rows,
                               double *x, double *y, mpcqp_info *info) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_step_host before mpcqp_setup");
    if (xref && xref_rows != 1 && xref_rows != h->L.N) return fail(MPCQP_ERR_ARG, "xref_rows must be 1 or Np+1");
    if (h->step_blank && (!x0 || !uminus1 || !xref)) return fail(MPCQP_ERR_STATE, "mpcqp_step_host: the handle was set up from raw q, l, u and holds no x0 / u_{-1} / xref yet: give all three");
    HIPCHK(hipSetDevice(h->device));
    const bool split = h->auto_balance && h->ncu > 0 && h->batch > h->ncu;
    if (split || ensure_pinned(h)) {
        int rc = mpcqp_update(h, x0, uminus1, xref, xref_rows);      // (clears step_blank once the step data is in)
        if (rc) return rc;
        if ((rc = mpcqp_solve(h))) return rc;
        return mpcqp_get_solution(h, x, y, info);
    }
    const Lay &L = h->L; const size_t B = (size_t)h->batch;
    h->L.raw = 0;
    if (xref) h->L.xref_rows = xref_rows;
    const int nxr = xref ? xref_rows * L.nx : 0;
    const bool inl = B == 1 && x0 && uminus1 && xref && L.nx + L.nu + nxr <= 32;      // (a single controller's few doubles travel in the kernel arguments instead)
    for (size_t b = 0; b < B && !inl; ++b) {
        double *dst = h->pin_in + b * h->pin_stride;
        if (x0) memcpy(dst, x0 + b * L.nx, sizeof(double) * L.nx);
        if (uminus1) memcpy(dst + L.nx, uminus1 + b * L.nu, sizeof(double) * L.nu);
        if (xref) memcpy(dst + L.nx + L.nu, xref + b * nxr, sizeof(double) * nxr);
    }
    RunArgs R; memset(&R, 0, sizeof(R));
    if (inl) {
        memcpy(R.inl, x0, sizeof(double) * L.nx); memcpy(R.inl + L.nx, uminus1, sizeof(double) * L.nu); memcpy(R.inl + L.nx + L.nu, xref, sizeof(double) * nxr);
        R.inl_n = L.nx + L.nu + nxr;
    }
    R.pin_in = (!inl && (x0 || uminus1 || xref)) ? (const double *)h->pin_in_dev : nullptr;
    R.pin_stride = h->pin_stride; R.pin_mask = (x0 ? 1 : 0) | (uminus1 ? 2 : 0) | (xref ? 4 : 0); R.pin_xref = nxr;
    R.pub = (double *)h->pin_out_dev; R.done = (unsigned *)h->npending_dev + 2; R.seq = ++h->host_seq;
    int rc = launch_run(h, R, 0);
    if (rc) return rc;
    h->step_blank = false;                                     // the step data has been accepted (a failed launch leaves the handle demanding it again)
    volatile unsigned long long *flag = (volatile unsigned long long *)((char *)h->pin_out + sizeof(double) * B * (L.n + L.m) + sizeof(mpcqp_info) * B);
    // Poll the flag: a short busy wait (a small solve is done within tens of microseconds), then yield the core between polls -- a long
    // solve (iteration limit, a long horizon) does not keep a CPU core spinning.
    for (unsigned long long spins = 0; *flag != h->host_seq; ++spins) {
        if (spins < 20000) cpu_relax(); else sched_yield();
        if ((spins & 0xfffff) == 0xfffff) {                    // every ~million polls: has the stream died?
            hipError_t e = hipStreamQuery(h->stream);
            if (e != hipSuccess && e != hipErrorNotReady) return fail(MPCQP_ERR_HIP, std::string("mpcqp_step_host: ") + hipGetErrorString(e));
            if (e == hipSuccess && *flag != h->host_seq) { HIPCHK(hipStreamSynchronize(h->stream)); break; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (x) memcpy(x, h->pin_out, sizeof(double) * B * L.n);
    if (y) memcpy(y, h->pin_out + B * L.n, sizeof(double) * B * L.m);
    if (info) memcpy(info, h->pin_out + B * (L.n + L.m), sizeof(mpcqp_info) * B);
    return MPCQP_OK;
}

extern "C" int mpcqp_get_stats(mpcqp_handle *h, uint64_t *out4, int reset) {
    if (!h || !out4) return fail(MPCQP_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out4, h->P.stats, 4 * sizeof(uint64_t), hipMemcpyDefault, h->stream));
#ifdef MPCQP_RUN_TIMING
    { uint64_t t[4]; hipMemcpy(t, h->P.stats + 4, sizeof(t), hipMemcpyDeviceToHost); fprintf(stderr, "phase wall-clock ticks: begin %llu admm %llu check %llu\n", (unsigned long long)t[0], (unsigned long long)t[1], (unsigned long long)t[2]);
      unsigned long long g[16]; hipMemcpyFromSymbol(g, HIP_SYMBOL(g_ticks), sizeof(g)); unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), z, sizeof(z));
      if (h->L.nw == 8) mpcqp_w8_ticks(g);      // (the 512-thread kernels count in their own translation unit)
      { double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)g[i]; if (tot <= 0) tot = 1;
        fprintf(stderr, "iteration cycles (thread 0, summed over workgroups): rhs %.1f%% fwd %.1f%% middle %.1f%% bwd %.1f%% t4 %.1f%% t5 %.1f%% t6 %.1f%% total %.3g;  outside the iterations (same unit): t7 %.3g t8 %.3g t9 %.3g;  check: setup+tail %.3g rows %.3g vars %.3g reduce %.3g decide %.3g\n",
                100 * g[0] / tot, 100 * g[1] / tot, 100 * g[2] / tot, 100 * g[3] / tot, 100 * g[4] / tot, 100 * g[5] / tot, 100 * g[6] / tot, tot, (double)g[7], (double)g[8], (double)g[9], (double)g[10], (double)g[11], (double)g[12], (double)g[13], (double)g[14]); } }
#endif
    if (reset) HIPCHK(hipMemsetAsync(h->P.stats, 0, 8 * sizeof(uint64_t), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_get_launch_times(mpcqp_handle *h, uint64_t *out, int nsteps) {
    if (!h || !out || nsteps < 0 || nsteps > TS_STEPS) return fail(MPCQP_ERR_ARG, "mpcqp_get_launch_times: bad argument (0 <= nsteps <= 64)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy2DAsync(out, sizeof(uint64_t) * (size_t)(2 + nsteps), h->P.tstamp, sizeof(uint64_t) * (size_t)TS_STRIDE, sizeof(uint64_t) * (size_t)(2 + nsteps),
                            (size_t)h->batch, hipMemcpyDefault, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_profile(mpcqp_handle *h, int enable, double *run_ms, int64_t *run_launches, int reset) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    int rc = drain_events(h);
    if (rc) return rc;
    if (enable >= 0) h->profiling = enable != 0;
    if (run_ms) *run_ms = h->run_ms;
    if (run_launches) *run_launches = h->run_launches;
    if (reset) { h->run_ms = 0.0; h->run_launches = 0; }
    return MPCQP_OK;
}

extern "C" int mpcqp_get_shape(mpcqp_handle *h, int *nx, int *nu, int *Np, int *Nc) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (nx) *nx = h->L.nx;
    if (nu) *nu = h->L.nu;
    if (Np) *Np = h->L.Np;
    if (Nc) *Nc = h->L.Nc;
    return MPCQP_OK;
}

extern "C" int mpcqp_get_dims(mpcqp_handle *h, int *n, int *m, int64_t *factor_doubles, int64_t *nnzL) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    const Lay &L = h->L;
    if (n) *n = L.n;
    if (m) *m = L.m;
    if (factor_doubles) *factor_doubles = h->P.fsz;
    if (nnzL) {      // structural nonzeros of the block factor (diagonal included) + the eliminated eps pivots
        int64_t nb = L.nb, nx = L.nx;
        const int64_t T = L.NcT;                       // stages with an input inside the tridiagonal part
        int64_t full = T * (nb * (nb + 1) / 2) + (int64_t)(L.N - T) * (nx * (nx + 1) / 2);
        int64_t sub = (T > 0 ? (T - 1) * (nx * nb + (int64_t)L.nu * L.nu) + nx * nb : 0) + (int64_t)(L.N - 1 - T) * nx * nx;
        int64_t border = L.border ? (int64_t)L.nu * (L.n_x + T * L.nu) + (int64_t)L.nu * (L.nu + 1) / 2 : 0;
        *nnzL = full + sub + border + (L.soft ? 2 * (int64_t)L.n_x : 0);
    }
    return MPCQP_OK;
}

// Bytes one instance moves between the memory system (HBM / Infinity Cache / L2) and its compute unit BY DESIGN of these
// kernels -- the roofline numerator bench.py uses ("Roofline accounting"):
//   per ADMM iteration : the factor stream (FactorFmt, mpcqp_factor.h).  16 x 16 stages: forward matrices of N-1 stages, packed
//                        S^-1 of N stages, N-1 stage tables and [G | G'], once each; 32 x 32 stages (S^-1-only): packed S^-1
//                        twice, one table per sweep, [G | G'] by each sweeping wave.  Problems whose iterate does not fit
//                        LDS also read and write x, z, y and read omega, s, q every iteration; Nc < Np adds the two border matrices.
//   per round (check)  : residual evaluation inputs (weights, D, E, omega, s, last increments), the round's load/store of
//                        the LDS-resident iterate and of the per-thread register copies of omega, s, q.
//   per solve          : the begin phase (step data, E, constraint types, q rebuild) and the solution write-out.
extern "C" int mpcqp_get_stream_bytes(mpcqp_handle *h, int64_t *per_iter, int64_t *per_round, int64_t *per_solve) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    const Lay &L = h->L;
    const int64_t n = L.n, m = L.m, nq = L.n_x + L.n_u, NB = L.NB;
    const int64_t sinv = L.fstage - L.ffwd - L.ftab;
    int64_t it = (L.dense || L.bcr) ? 0 /* the factor sits in registers for the round */ : !L.ffwd ? 2 * (int64_t)L.N * sinv + (int64_t)L.N * L.ftab + 2 * (int64_t)L.fhead   // S^-1-only build: S^-1 twice, one of the two tables per sweep, [G | G'] by each sweeping wave
                         : (int64_t)(L.N - 1) * L.ffwd + (int64_t)L.N * sinv + (int64_t)(L.N - 1) * L.ftab + L.fhead;   // forward matrices once, S^-1 once, the tables, G / G' once each
    if (L.NB >= 64) it = (3 * (int64_t)L.N - 2) * (int64_t)L.NB * L.NB;      // wide stages: S^-1 of every stage, M and M' of all but the last
    if (L.grp) it = (3 * (int64_t)group_count(L.N, L.grp) - 1) * GroupFmt::NN;      // grouped small stages: S^-1 of every super-stage, forward matrix and its transpose of all but the ends, the middle's second pair
    if (!h->lds_state && !L.lstage) it += (2 * n + 4 * m) /* x, z, y read + written */ + (m + L.n_x) /* omega */ + (n + L.n_x) /* s */ + nq /* q */;
    if (L.border && !L.dense) it += 2 * (int64_t)L.nu * L.N * NB;      // (the dense inverse holds the held input's couplings itself)
    int64_t rd = (L.model_sz - L.hot_lds) /* the weight matrices, unless they are staged with the hot prefix */ + 2 * n + 2 * m /* D, s, E, omega */ + n + m /* dx, dy */;
    if (L.dense) rd += DenseFmt::DOUBLES;               // the round's load of K^-1 into registers
    if (L.bcr && !L.bcrtop) rd += (int64_t)L.bcr * BcrFmt::REC - 4 * BcrFmt::NN;      // ... of the cyclic-reduction fragments (the two end stages have one neighbour)
    int64_t fr = 0;
    if (L.bcrtop) {                                     // ... of levels 0 and 1 (the top inverse and G, G' enter LDS once per kernel prologue: admm_latw)
        for (int l = 0; l < 2; ++l) for (int kind = 0; kind < 3; ++kind) for (int t = 0; t < lat_count(L.bcr, l, kind); ++t) fr += lat_nfr(L.bcr, l, kind, t);
        rd += fr * BcrFmt::NN;
    }
    rd += (h->lds_state || L.lstage) ? 2 * (n + 2 * m) /* iterate in and out of LDS */ + (m + L.n_x) + (n + L.n_x) + nq : (n + 2 * m);
    if (L.lstage && L.border) rd += 2 * (int64_t)L.nu * L.N * NB;      // the border matrices staged with it
    int64_t sv = L.hot_lds + L.step_sz + 3 * m /* E, types, omega */ + nq + 2 * (n + m) /* solution out, iterate read */;
    if (L.bcrtop && L.nw == 8 && (L.bcr + 3) / 4 <= 8 && L.hot_lds > L.hot_sz) {
        // The 512-thread latency round with its own termination test (latw_check): a round boundary moves the level fragments, the increments (written by
        // the last iteration, read back by the test) and the three scalings per lane the test clips the certificates with; the owners' registers are
        // loaded once per SOLVE (iterate, metric, linear cost), the iterate, the solution and the record written once per solve; the top inverse and the hot
        // prefix enter LDS once per kernel prologue -- per queue item of a persistent closed-loop launch, about every fifth solve (QUEUE_ITEMS_PER_SLOT).
        // (Spill traffic around the non-inlined test -- ~ 110 bytes per lane and round in the write counter -- is NOT design traffic and not in here.)
        rd = fr * BcrFmt::NN + 2 * (n + m) /* dx, dy out and back */ + 3 * 64 * 8 /* E of the owned rows */;
        sv = L.step_sz + 3 * m + nq                                            /* begin: E, types, omega looked at, q written */
           + (2 * n + 3 * m + nq)                                              /* the round's prologue: x, s, q, omega, z, y */
           + (n + 2 * m) + (n + m)                                             /* the solve's end: iterate, reported solution */
           + ((int64_t)L.bcrtop * L.bcrtop * BcrFmt::NN + L.hot_lds) / 5;     /* kernel prologue, amortised */
    }
    if (per_iter) *per_iter = 8 * it;
    if (per_round) *per_round = 8 * rd;
    if (per_solve) *per_solve = 8 * sv;
    return MPCQP_OK;
}

// Matrix-core instructions (v_mfma_f64_4x4x4_4b_f64, 512 flop each, a quarter of them useful in a mat-vec) one instance issues per
// ADMM iteration with this handle's backend -- the numerator of the MFMA roofline bench.py reports for the latency backend.
extern "C" int mpcqp_get_work(mpcqp_handle *h, int64_t *mfma_per_iter) {
    if (!h || !mfma_per_iter) return fail(MPCQP_ERR_ARG, "null argument");
    const Lay &L = h->L;
    int64_t mv = 0;                                  // 16 x 16 mat-vecs (four MFMAs each)
    if (L.dense || L.NB >= 64) mv = 0;               // vector ALU only
    else if (L.bcr) {
        for (int l = 0; l < (L.bcrtop ? 2 : bcr_levels(L.bcr)); ++l)
            for (int kind = 0; kind < 3; ++kind) for (int t = 0; t < lat_count(L.bcr, l, kind); ++t) mv += lat_nfr(L.bcr, l, kind, t);
        if (!LATW_TOP_VALU) mv += (int64_t)L.bcrtop * L.bcrtop;      // dense top on the matrix cores: one mat-vec per block of the inverse (LATW_TOP_VALU: 512 nt^2 flop per iteration on the vector ALU instead, not counted here)
        mv += 2 * ((L.bcr + 3) / 4);                 // G v and G'W: one group per four stages each
    } else if (L.grp) mv = 3 * (int64_t)group_count(L.N, L.grp) - 1;      // forward 1, backward 2 mat-vecs per super-stage, the middle stage
    else {
        const int64_t blk = (L.NB / 16) * (L.NB / 16);
        mv = L.ffwd ? (int64_t)(L.N - 1) * blk * 3 + 3 * blk : (int64_t)(L.N - 1) * blk * 4 + 3 * blk;      // forward 1 (or 2) + backward 2 mat-vecs per stage, the middle stage
    }
    *mfma_per_iter = 4 * mv;
    return MPCQP_OK;
}

// How many workgroups of the handle's solve kernel one compute unit holds at a time (the kernels' __launch_bounds__ occupancy, run_occupancy in
// mpcqp_kernels.h, capped by what the dynamic LDS block leaves room for) and the compute units of the handle's device: bench.py's "instances in
// flight" (what the memory-side cache sees) = the product.
extern "C" int mpcqp_get_occupancy(mpcqp_handle *h, int *workgroups_per_cu, int *compute_units, int *threads_per_workgroup) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    const Lay &L = h->L;
    const bool one_at_a_time = L.bcr || L.dense || L.NB > 32 || L.nw == 8;
    int occ = one_at_a_time ? 1 : (L.NB <= 16 ? 4 : 2);
    const size_t lds_cu = 160 * 1024;
    if (h->smem_solve > 0) occ = std::max(1, std::min(occ, (int)(lds_cu / h->smem_solve)));
    if (workgroups_per_cu) *workgroups_per_cu = occ;
    if (compute_units) *compute_units = h->ncu;
    if (threads_per_workgroup) *threads_per_workgroup = L.nw == 8 ? 512 : 256;
    return MPCQP_OK;
}

// Name of the k_mpc_run instantiation the handle's solves (loop = 0) or closed-loop runs (loop = 1) launch, spelled as
// rocprofv3 prints it (without spaces) -- so that bench.py and the profile summaries name the same kernel.
extern "C" int mpcqp_kernel_name(mpcqp_handle *h, int loop, char *buf, int buflen) {
    if (!h || !buf || buflen < 1) return fail(MPCQP_ERR_ARG, "null argument");
    const Lay &L = h->L;
    const bool spec = L.dense ? false : (L.NB == 16 && !L.bcr && !h->lds_state && L.nx == 4 && L.nu == 1) ? true :
                      !L.border && (L.bcr ? (L.bcr == 31 && L.nx == 12 && L.nu == 4)
                                          : ((L.NB == 16 && h->lds_state && L.nx == 12 && L.nu == 4) || (L.NB == 32 && !h->lds_state && L.nx == 20 && L.nu == 8)));
    snprintf(buf, (size_t)buflen, "%sk_mpc_run<%d,%s,%d,%d,%d,%s>", L.nw == 8 ? "w8::" : "", L.NB, h->lds_state ? "true" : "false", spec ? L.nx : 0, spec ? L.nu : 0,
             L.dense ? MODE_DENSE : L.bcr ? (L.bcrtop ? MODE_BCRT : MODE_BCR) + L.bcr : L.border ? MODE_BORDER : MODE_CHAIN, loop ? "true" : "false");
    return MPCQP_OK;
}

extern "C" int mpcqp_warm_start(mpcqp_handle *h, const double *x, const double *y) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "warm_start before setup");
    HIPCHK(hipSetDevice(h->device));
    size_t B = (size_t)h->batch;
    if (x) { HIPCHK(hipMemcpyAsync(h->P.x, x, B * h->L.n * sizeof(double), hipMemcpyDefault, h->stream)); h->warm_x_pending = true; }
    if (y) { HIPCHK(hipMemcpyAsync(h->P.y, y, B * h->L.m * sizeof(double), hipMemcpyDefault, h->stream)); }
    return MPCQP_OK;
}

extern "C" int mpcqp_export_qp(mpcqp_handle *h, double *Pm, double *Am, double *q, double *l, double *u) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "export before setup");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; size_t B = (size_t)h->batch;
    double *dP = nullptr, *dA = nullptr, *dq = nullptr, *dl = nullptr, *du = nullptr;
    Scratch sc;
    if (Pm) { HIPCHK(sc.get(&dP, B * L.n * L.n)); HIPCHK(hipMemsetAsync(dP, 0, B * L.n * L.n * sizeof(double), h->stream)); }
    if (Am) { HIPCHK(sc.get(&dA, B * L.m * L.n)); HIPCHK(hipMemsetAsync(dA, 0, B * L.m * L.n * sizeof(double), h->stream)); }
    if (q) HIPCHK(sc.get(&dq, B * L.n));
    if (l && u) { HIPCHK(sc.get(&dl, B * L.m)); HIPCHK(sc.get(&du, B * L.m)); }
    DISPATCH_NB(L.NB, {
        if (set_smem(k_export<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_export<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, dP, dA, dq, dl, du);
    });
    HIPCHK(hipGetLastError());
    int rc = 0;
    rc |= get(h, Pm, dP, B * L.n * L.n * sizeof(double)); rc |= get(h, Am, dA, B * L.m * L.n * sizeof(double));
    rc |= get(h, q, dq, B * L.n * sizeof(double));
    if (l && u) { rc |= get(h, l, dl, B * L.m * sizeof(double)); rc |= get(h, u, du, B * L.m * sizeof(double)); }
    HIPCHK(hipStreamSynchronize(h->stream));
    return rc ? MPCQP_ERR_HIP : MPCQP_OK;
}

extern "C" int mpcqp_get_scaling(mpcqp_handle *h, double *D, double *E, double *c, double *rho) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    size_t B = (size_t)h->batch;
    if (get(h, D, h->P.D, B * h->L.n * sizeof(double)) || get(h, E, h->P.E, B * h->L.m * sizeof(double)) ||
        get(h, c, h->P.c, B * sizeof(double)) || get(h, rho, h->P.rho, B * sizeof(double))) return MPCQP_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_get_iterate(mpcqp_handle *h, double *x, double *z, double *y) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    size_t B = (size_t)h->batch;
    if (get(h, x, h->P.x, B * h->L.n * sizeof(double)) || get(h, z, h->P.z, B * h->L.m * sizeof(double)) ||
        get(h, y, h->P.y, B * h->L.m * sizeof(double))) return MPCQP_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_debug_kkt_solve(mpcqp_handle *h, const double *rhs, double *sol) {
    if (!h || !rhs || !sol) return fail(MPCQP_ERR_ARG, "null argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "kkt_solve before setup");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; size_t bytes = (size_t)h->batch * L.n * sizeof(double);
    double *dr = nullptr, *ds = nullptr;
    Scratch sc;
    HIPCHK(sc.get(&dr, (size_t)h->batch * L.n)); HIPCHK(sc.get(&ds, (size_t)h->batch * L.n));
    HIPCHK(hipMemcpyAsync(dr, rhs, bytes, hipMemcpyDefault, h->stream));
    DISPATCH_NB(L.NB, {
        if (set_smem(k_kkt_solve<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_kkt_solve<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, dr, ds);
    });
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sol, ds, bytes, hipMemcpyDefault, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

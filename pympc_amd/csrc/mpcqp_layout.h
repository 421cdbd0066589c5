// mpcqp_layout.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Instance layout, device pointer tables, visitor context.
#pragma once

// ------------------------------------------------------------------------------------------------
// Layout of one instance
// ------------------------------------------------------------------------------------------------
constexpr int MAXEV = 64;         // profiling: launches whose HIP events may be pending
constexpr int TS_STEPS = 64, TS_STRIDE = 2 + TS_STEPS;      // per-instance time stamps of a closed-loop launch (Ptrs::tstamp)

struct Lay {
    int nx, nu, Np, Nc, N, nb, n, m, n_x, n_u, ou, oe, rs, ri, rdu;
    int NB;                       // padded stage block size (16, 32, 64 or 128)
    int soft;                     // 1: slack columns eps_k (soft state box, pyMPC's SOFT_ON); 0: none, the state box is hard
    int NcT;                      // stages 0..NcT-1 carry their input u_k inside the block-tridiagonal part
    int border;                   // 1 if Nc < Np: the held last input u_{Nc-1} couples to every later stage and is
                                  // handled as a bordered (Schur-complement) correction, see border_* below
    float rnx, rnu;               // reciprocals for cheap index division
    // model blob: hot prefix [Ad|Bd|xmin|xmax|umin|umax|Dumin|Dumax|uref|eps_feas] then [Qx|QxN|Qu|QDu]
    int oAd, oBd, oxmin, oxmax, oumin, oumax, oDumin, oDumax, ouref, oeps, hot_sz;
    int oQx, oQxN, oQu, oQDu, model_sz;
    int hot_lds;                  // doubles of the model blob every kernel stages into LDS at its start: hot_sz, or model_sz where the workgroup's LDS has room
                                  // for the weight matrices too (mpcqp_create: kernels that run one or two workgroups per compute unit) -- a termination check
                                  // then reads them where they are instead of copying them in (a global round trip and a barrier per check)
    int step_sz;                  // [x0 | um1 | xref(N*nx) | du0lo(nu) | du0hi(nu)]  (the last two: raw-vector mode only)
    int odu0;                     // offset of du0lo in the step blob
    int raw;                      // 1: q and the bounds are what mpcqp_update_vectors uploaded (not rebuilt from x0, u_{-1}, xref)
    int xref_rows;                // 1 or N
    int fstage;                   // doubles per factor stage (FactorFmt<NB>::STAGE): [forward matrix | packed S^-1 | table]
    int fhead, ffwd, ftab;        // doubles of the per-instance factor header ([G | G']) / of a stage's forward matrix / of its off-diagonal table (hybrid)
    int tsz;                      // LDS work vector length: max(m, 4*NB*NB)
    int dense;                    // 1: small problem -- the KKT solve is a dense mat-vec with K^-1 held in registers (mpcqp_dense.h)
    int bcr;                      // > 0: block cyclic reduction instead of the twisted sweeps (mpcqp_bcr.h; small batches of 16 x 16 stages); the value is the
                                  // stage count of the schedule the handle runs (11, 21 or 31 >= N: stages N .. bcr-1 are identity padding)
    int bcrtop;                   // > 0 (with bcr): the cyclic reduction stops after two levels and the stages that are left -- this many, every fourth -- are
                                  // solved with the explicit inverse of their Schur complement (BcrFmt: the fragments behind the stage records; mpcqp_latw.h)
    int nw;                       // waves per workgroup of the handle's solve kernels: 4 (NT = 256), or 8 for the kernels of mpcqp_w8.hip (sizes the reduction scratch)
    int lstage;                   // 1: the iterate lives in global memory between rounds but is STAGED into LDS for a round (admm_round_global: small batches of problems too large for the owner-mapped LDS round)
    int grp;                      // > 1: this many consecutive small stages share a 16 x 16 block of the KKT factor (mpcqp_group.h; long horizons of small stages)
    int NR, dld;                  // dense: unknowns N*(nx+nu) of the reduced KKT system; LDS row stride of the inversion workspace (odd)
};

struct Ptrs {
    double *model, *step;
    double *D, *E, *c, *omega, *s, *rho;
    double *F;
    double *x, *z, *y;            // iterate (unscaled units)
    double *xo, *yo;              // reported solution
    double *dx, *dy;              // last primal / dual increments (infeasibility certificates)
    double *Bb, *Zb, *Sig;        // border (Nc < Np): K[:,ubar] and T^-1 K[:,ubar] in padded layout [nu][N*NB], Schur inverse [nu*nu]
    double *bws;                  // global workspace of a factorization: block cyclic reduction [batch][N * BcrFmt::WSTAGE]; 128-wide stages [batch][HugeFmt::GWS]
    double *qv;                   // linear cost of the x,u variables [n_x+n_u] (rebuilt by every kernel prologue)
    double *Dt, *Et;              // Ruiz temporaries
    int *ctype;
    unsigned long long *stats;    // [0] ADMM iterations, [1] residual evaluations, [2] refactorizations, [3] instance-solves
    mpcqp_info *info;
    const int *perm;              // workgroup -> instance map (load balancing, see rebalance() in mpcqp.hip), or null = identity
    unsigned *work;               // per instance: ADMM iterations since the map was last rebuilt
    unsigned long long *tstamp;   // per instance [TS_STRIDE]: { entry, exit, end of step 0 .. TS_STEPS-1 } of the last closed-loop launch, 100 MHz ticks (mpcqp_get_launch_times)
    long long fsz;                // factor doubles per instance
    int *fown;                    // null, or (mpcqp_share_factor) per instance the factor slot it SOLVES with: its own (b) or the shared one (slot `batch`, a copy of
                                  // instance 0's factor that no kernel writes); an instance that refactors writes its own slot and points itself back at it
};

// The hot kernel gets only the pointers it uses (fewer scalar registers -> no SGPR spills into vector lanes).
struct HotPtrs {
    const double *model, *step, *omega, *s, *qv, *F, *c, *Bb, *Zb, *Sig;
    double *x, *z, *y, *dx, *dy;
    const int *perm;
    long long fsz;
    const int *fown;
};

// The factor an instance's KKT solves read (streaming backends): its own slot, or the shared one (Ptrs::fown).  The map entry is read past the
// vector L1 (another workgroup may have handled the instance's previous step range of the same launch and un-shared it there).
template <class PT>
__device__ __forceinline__ const double *factor_of(const PT &P, int b) {
    long long slot = b;
    if (P.fown) slot = __builtin_amdgcn_readfirstlane(__hip_atomic_load(P.fown + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return P.F + (size_t)slot * P.fsz;
}
// An instance that is about to rewrite its factor (rho update, changed constraint types, mpcqp_refactor) solves with its own slot from then on.
template <class PT>
__device__ __forceinline__ void factor_unshare(const PT &P, int b) {
    if (P.fown && threadIdx.x == 0) { __hip_atomic_store(P.fown + b, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __threadfence(); }
}

// The instance this workgroup works on.  (Which workgroup handles which instance never changes a result; the map
// only decides which instances share a CU.)
// (The map's upper byte carries the instance's PACE: how many s_sleep(127) units -- 3.5 us each -- it idles per ADMM iteration in the co-resident
//  bandwidth kernels, see rebalance() in mpcqp.hip: instances expected to need few iterations give way to the stragglers the launch waits for.)
constexpr int PERM_INST_MASK = 0x00FFFFFF, PERM_PACE_SHIFT = 24;
__device__ __forceinline__ int inst_of(const int *perm) { return perm ? (perm[blockIdx.x] & PERM_INST_MASK) : (int)blockIdx.x; }
__device__ __forceinline__ int pace_of(const int *perm) { return perm ? (int)((unsigned)perm[blockIdx.x] >> PERM_PACE_SHIFT) : 0; }

__device__ __forceinline__ int idiv(int r, float rcp) { return __float2int_rd(((float)r + 0.5f) * rcp); }
__device__ __forceinline__ double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

// Everything a row visitor needs.  `hot` points at the hot prefix of the model blob (LDS or global),
// `Q` at the blob itself (global) for the weight matrices.
struct Ctx {
    Lay L;
    const double *hot;
    const double *wts;            // the weight matrices [Qx|QxN|Qu|QDu] (global model blob + hot_sz, or a copy in LDS)
    __device__ __forceinline__ const double *Ad() const { return hot + L.oAd; }
    __device__ __forceinline__ const double *Bd() const { return hot + L.oBd; }
    __device__ __forceinline__ const double *Qx() const { return wts + (L.oQx - L.hot_sz); }
    __device__ __forceinline__ const double *QxN() const { return wts + (L.oQxN - L.hot_sz); }
    __device__ __forceinline__ const double *Qu() const { return wts + (L.oQu - L.hot_sz); }
    __device__ __forceinline__ const double *QDu() const { return wts + (L.oQDu - L.hot_sz); }
    __device__ __forceinline__ double eps_feas() const { return hot[L.oeps]; }
};

#ifdef MPCQP_RUN_TIMING
// Development build: shader-clock cycles between phase boundaries as seen by thread 0, summed in LDS and flushed to g_ticks once
// per ADMM phase (an atomic per tick would perturb what it measures).
__device__ unsigned long long g_ticks[16];
__device__ __forceinline__ unsigned long long *tick_slots() { __shared__ unsigned long long s[16]; return s; }
#define TICK(i) { if (threadIdx.x == 0) { unsigned long long *s_ = tick_slots(); const unsigned long long t_ = clock64(); s_[i] += t_ - s_[15]; s_[15] = t_; } }
#define TICK_START { if (threadIdx.x == 0) tick_slots()[15] = clock64(); }
#define TICK_RESET { if (threadIdx.x < 16) tick_slots()[threadIdx.x] = 0; __syncthreads(); }
#define TICK_FLUSH { __syncthreads(); if (threadIdx.x < 15) atomicAdd(&g_ticks[threadIdx.x], tick_slots()[threadIdx.x]); }
#else
#define TICK(i)
#define TICK_START
#define TICK_RESET
#define TICK_FLUSH
#endif

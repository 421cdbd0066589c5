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
//   - one 256-thread workgroup owns one instance for the whole solve: iterate in LDS, wave 0 runs the
//     sequential block forward/backward sweeps on the matrix cores (v_mfma_f64_4x4x4_4b_f64, stage output
//     registers = next stage's B operand), all waves run the stage-parallel parts.
//
// FP64 throughout.  No CPU fallback exists in this library.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"

#ifndef NT
#define NT 256                 // threads per workgroup (one workgroup = one MPC instance)
#endif
#define NWAVES (NT / 64)
#define QP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(MPCQP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------------
// Layout of one instance
// ------------------------------------------------------------------------------------------------
constexpr int MAXEV = 64;         // profiling: launches whose HIP events may be pending

struct Lay {
    int nx, nu, Np, Nc, N, nb, n, m, n_x, n_u, ou, oe, rs, ri, rdu;
    int NB;                       // padded stage block size (16 or 32)
    int NcT;                      // stages 0..NcT-1 carry their input u_k inside the block-tridiagonal part
    int border;                   // 1 if Nc < Np: the held last input u_{Nc-1} couples to every later stage and is
                                  // handled as a bordered (Schur-complement) correction, see border_* below
    float rnx, rnu;               // reciprocals for cheap index division
    // model blob: hot prefix [Ad|Bd|xmin|xmax|umin|umax|Dumin|Dumax|uref|eps_feas] then [Qx|QxN|Qu|QDu]
    int oAd, oBd, oxmin, oxmax, oumin, oumax, oDumin, oDumax, ouref, oeps, hot_sz;
    int oQx, oQxN, oQu, oQDu, model_sz;
    int step_sz;                  // [x0 | um1 | xref(N*nx)]
    int xref_rows;                // 1 or N
    int fstage;                   // doubles per factor stage: 2*NB*NB  [forward matrix | S^-1]
    int tsz;                      // LDS work vector length: max(m, 4*NB*NB)
};

struct Ptrs {
    double *model, *step;
    double *D, *E, *c, *omega, *s, *rho;
    double *F;
    double *x, *z, *y;            // iterate (unscaled units)
    double *xo, *yo;              // reported solution
    double *dx, *dy;              // last primal / dual increments (infeasibility certificates)
    double *Bb, *Zb, *Sig;        // border (Nc < Np): K[:,ubar] and T^-1 K[:,ubar] in padded layout [nu][N*NB], Schur inverse [nu*nu]
    double *qv;                   // linear cost of the x,u variables [n_x+n_u] (rebuilt by every kernel prologue)
    double *Dt, *Et;              // Ruiz temporaries
    int *ctype;
    unsigned long long *stats;    // [0] ADMM iterations, [1] residual evaluations, [2] refactorizations, [3] instance-solves
    mpcqp_info *info;
    long long fsz;                // factor doubles per instance
};

// The hot kernel gets only the pointers it uses (fewer scalar registers -> no SGPR spills into vector lanes).
struct HotPtrs {
    const double *model, *step, *omega, *s, *qv, *F, *c, *Bb, *Zb, *Sig;
    double *x, *z, *y, *dx, *dy;
    long long fsz;
};

__device__ __forceinline__ int idiv(int r, float rcp) { return __float2int_rd(((float)r + 0.5f) * rcp); }
__device__ __forceinline__ double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

// Everything a row visitor needs.  `hot` points at the hot prefix of the model blob (LDS or global),
// `Q` at the blob itself (global) for the weight matrices.
struct Ctx {
    Lay L;
    const double *hot;
    const double *blob;
    __device__ __forceinline__ const double *Ad() const { return hot + L.oAd; }
    __device__ __forceinline__ const double *Bd() const { return hot + L.oBd; }
    __device__ __forceinline__ const double *Qx() const { return blob + L.oQx; }
    __device__ __forceinline__ const double *QxN() const { return blob + L.oQxN; }
    __device__ __forceinline__ const double *Qu() const { return blob + L.oQu; }
    __device__ __forceinline__ const double *QDu() const { return blob + L.oQDu; }
    __device__ __forceinline__ double eps_feas() const { return hot[L.oeps]; }
};

// ------------------------------------------------------------------------------------------------
// Row visitors: enumerate (coefficient, index) of one row of A, one column of A, one row of P.
// They ARE the device-side definition of the reference's matrices (mpc.py:482-598).
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void A_row(const Ctx &c, int r, F f) {
    const Lay &L = c.L;
    if (r < L.rs) {                                   // dynamics rows  (mpc.py:537-552)
        int k = idiv(r, L.rnx), i = r - k * L.nx;
        f(-1.0, r);
        if (k > 0) {
            const double *a = c.Ad() + i * L.nx;
            int base = (k - 1) * L.nx;
            for (int j = 0; j < L.nx; ++j) f(a[j], base + j);
            int ku = min(k - 1, L.Nc - 1);
            const double *b = c.Bd() + i * L.nu;
            base = L.ou + ku * L.nu;
            for (int j = 0; j < L.nu; ++j) f(b[j], base + j);
        }
    } else if (r < L.ri) {                            // soft state box: x_k + eps_k  (mpc.py:555-559)
        int j = r - L.rs;
        f(1.0, j);
        f(1.0, L.oe + j);
    } else if (r < L.rdu) {                           // input box  (mpc.py:561-565)
        f(1.0, L.ou + (r - L.ri));
    } else {                                          // Delta-u rows  (mpc.py:569-580)
        int rr = r - L.rdu;
        if (rr < L.nu) f(1.0, L.ou + rr);
        else {
            int cc = rr - L.nu;                       // -I + superdiagonal at offset ONE SCALAR
            f(-1.0, L.ou + cc);
            if (cc + 1 < L.n_u) f(1.0, L.ou + cc + 1);
        }
    }
}

template <class F>
__device__ __forceinline__ void AT_row(const Ctx &c, int j, F f) {     // column j of A
    const Lay &L = c.L;
    if (j < L.ou) {
        int k = idiv(j, L.rnx), i = j - k * L.nx;
        f(-1.0, j);
        if (k < L.Np) {
            const double *a = c.Ad() + i;
            int base = (k + 1) * L.nx;
            for (int r = 0; r < L.nx; ++r) f(a[r * L.nx], base + r);
        }
        f(1.0, L.rs + j);
    } else if (j < L.oe) {
        int cc = j - L.ou;
        int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
        int s_end = (k == L.Nc - 1) ? L.Np : k + 1;   // the last input is held to the end of the horizon
        const double *b = c.Bd() + jj;
        for (int s = k + 1; s <= s_end; ++s) {
            int base = s * L.nx;
            for (int r = 0; r < L.nx; ++r) f(b[r * L.nu], base + r);
        }
        f(1.0, L.ri + cc);
        if (k == 0) f(1.0, L.rdu + jj);
        f(-1.0, L.rdu + L.nu + cc);
        if (cc > 0) f(1.0, L.rdu + L.nu + cc - 1);
    } else {
        f(1.0, L.rs + (j - L.oe));
    }
}

// P as the solver sees it: the upper triangle of the reference's P mirrored (osqp keeps triu(P)).
template <class F>
__device__ __forceinline__ void P_row(const Ctx &c, int j, F f) {
    const Lay &L = c.L;
    if (j < L.ou) {                                    // blkdiag(I (x) Qx, QxN)  (mpc.py:486-487)
        int k = idiv(j, L.rnx), i = j - k * L.nx;
        const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
        int base = k * L.nx;
        for (int l = 0; l < L.nx; ++l) f(Q[min(i, l) * L.nx + max(i, l)], base + l);
    } else if (j < L.oe) {                             // diag(iU) (x) Qu + iDu (x) QDu  (mpc.py:505-526)
        int cc = j - L.ou;
        int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
        double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
        double dk = (k == L.Nc - 1) ? 1.0 : 2.0;
        const double *Qu = c.Qu(), *QDu = c.QDu();
        int base = L.ou + k * L.nu;
        for (int l = 0; l < L.nu; ++l) {
            int a = min(jj, l), b = max(jj, l);
            f(__dadd_rn(__dmul_rn(iu, Qu[a * L.nu + b]), __dmul_rn(dk, QDu[a * L.nu + b])), base + l);
        }
        if (k + 1 < L.Nc) for (int l = 0; l < L.nu; ++l) f(-QDu[jj * L.nu + l], base + L.nu + l);
        if (k > 0) for (int l = 0; l < L.nu; ++l) f(-QDu[l * L.nu + jj], base - L.nu + l);
    } else {
        f(c.eps_feas(), j);                            // I (x) Qeps  (mpc.py:531)
    }
}

// Bounds of row r exactly as mpc.py:551-580 / 404-408 build them, clipped to +-1e30 like osqp's wrapper.
// x0s/um1s: current x0 and u_{-1}.
__device__ __forceinline__ void row_bounds(const Ctx &c, const double *x0s, const double *um1s, int r, double &lo, double &hi) {
    const Lay &L = c.L;
    if (r < L.rs) {
        lo = hi = (r < L.nx) ? -x0s[r] : 0.0;
    } else if (r < L.ri) {
        int j = r - L.rs; int k = idiv(j, L.rnx); int i = j - k * L.nx;
        lo = c.hot[L.oxmin + i]; hi = c.hot[L.oxmax + i];
    } else if (r < L.rdu) {
        int cc = r - L.ri; int k = idiv(cc, L.rnu); int jj = cc - k * L.nu;
        lo = c.hot[L.oumin + jj]; hi = c.hot[L.oumax + jj];
    } else {
        int rr = r - L.rdu; int k = idiv(rr, L.rnu); int jj = rr - k * L.nu;
        lo = c.hot[L.oDumin + jj]; hi = c.hot[L.oDumax + jj];
        if (rr < L.nu) { lo += um1s[jj]; hi += um1s[jj]; }
    }
    lo = lo < -QP_INFTY ? -QP_INFTY : lo;
    hi = hi > QP_INFTY ? QP_INFTY : hi;
}

// Linear cost of the x and u variables (eps part is zero): mpc.py:489-526 / 411-452.
__device__ void build_q(const Ctx &c, const double *step, double *Qv) {
    const Lay &L = c.L;
    const double *um1 = step + L.nx, *xref = step + L.nx + L.nu;
    const double *uref = c.hot + L.ouref;
    for (int j = threadIdx.x; j < L.n_x + L.n_u; j += NT) {
        double acc = 0.0;
        if (j < L.n_x) {
            int k = idiv(j, L.rnx), i = j - k * L.nx;
            const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
            if (L.xref_rows == 1) { for (int l = 0; l < L.nx; ++l) acc += Q[i * L.nx + l] * xref[l]; }
            else { const double *xr = xref + k * L.nx; for (int l = 0; l < L.nx; ++l) acc += xr[l] * Q[l * L.nx + i]; }
            acc = -acc;
        } else {
            int cc = j - L.n_x; int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
            double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
            double a = 0.0;
            for (int l = 0; l < L.nu; ++l) a += c.Qu()[jj * L.nu + l] * uref[l];
            acc = iu * (-a);
            if (k == 0) { double d = 0.0; for (int l = 0; l < L.nu; ++l) d += c.QDu()[jj * L.nu + l] * um1[l]; acc += -d; }
        }
        Qv[j] = acc;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Block reductions (4 waves of 64).  red: LDS scratch of >= 4*K doubles.
// ------------------------------------------------------------------------------------------------
template <int KMAX, int KSUM>
__device__ void block_reduce(double *vmax, double *vsum, double *red) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int K = KMAX + KSUM;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { double v = vmax[i]; for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); vmax[i] = v; }
#pragma unroll
    for (int i = 0; i < KSUM; ++i) { double v = vsum[i]; for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); vsum[i] = v; }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) red[wv * K + i] = vmax[i];
#pragma unroll
        for (int i = 0; i < KSUM; ++i) red[wv * K + KMAX + i] = vsum[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { double v = red[i]; for (int w = 1; w < NWAVES; ++w) v = fmax(v, red[w * K + i]); vmax[i] = v; }
#pragma unroll
    for (int i = 0; i < KSUM; ++i) { double v = red[KMAX + i]; for (int w = 1; w < NWAVES; ++w) v += red[w * K + KMAX + i]; vsum[i] = v; }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Reduced KKT matrix  K = c P + diag(s) + A' diag(omega) A  with eps eliminated: stage blocks.
// Stage k holds v_k = (x_k, u_k) (u absent in the last stage); blocks are NB x NB, identity padded.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double kkt_diag_entry(const Ctx &c, const double *om, const double *sv, double cc, int k, int a, int b) {
    const Lay &L = c.L;
    const int nbk = (k < L.NcT) ? L.nb : L.nx;
    if (a >= nbk || b >= nbk) return a == b ? 1.0 : 0.0;
    const double *Ad = c.Ad(), *Bd = c.Bd();
    const double *omd = om + (k + 1) * L.nx;          // dynamics rows of stage k+1
    double v = 0.0;
    if (a < L.nx && b < L.nx) {
        const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
        v = cc * Q[min(a, b) * L.nx + max(a, b)];
        if (k < L.Np) for (int r = 0; r < L.nx; ++r) v += Ad[r * L.nx + a] * omd[r] * Ad[r * L.nx + b];
        if (a == b) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e], se = sv[L.oe + e];
            double ce = cc * c.eps_feas() + se;
            v += sv[e] + om[e] + ws * (ce / (ce + ws));     // soft row with eps eliminated
        }
    } else if (a >= L.nx && b >= L.nx) {
        int ja = a - L.nx, jb = b - L.nx;
        double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
        double dk = (k == L.Nc - 1) ? 1.0 : 2.0;
        int lo = min(ja, jb), hi = max(ja, jb);
        v = cc * (iu * c.Qu()[lo * L.nu + hi] + dk * c.QDu()[lo * L.nu + hi]);
        for (int r = 0; r < L.nx; ++r) v += Bd[r * L.nu + ja] * omd[r] * Bd[r * L.nu + jb];
        int ca = k * L.nu + ja;
        const double *omdiff = om + L.rdu + L.nu;
        if (ja == jb) {
            v += sv[L.ou + ca] + om[L.ri + ca] + omdiff[ca];
            if (k == 0) v += om[L.rdu + ja];
            if (ca > 0) v += omdiff[ca - 1];
        } else if (hi - lo == 1) {
            v -= omdiff[k * L.nu + lo];
        }
    } else {
        int xa = a < L.nx ? a : b, ju = (a < L.nx ? b : a) - L.nx;
        for (int r = 0; r < L.nx; ++r) v += Ad[r * L.nx + xa] * omd[r] * Bd[r * L.nu + ju];
    }
    return v;
}

// K_{k+1,k}: rows = variables of stage k+1, cols = variables of stage k.
__device__ __forceinline__ double kkt_sub_entry(const Ctx &c, const double *om, double cc, int k, int a, int b) {
    const Lay &L = c.L;
    const int nbk = (k < L.NcT) ? L.nb : L.nx;
    const int nbn = (k + 1 < L.NcT) ? L.nb : L.nx;
    if (a >= nbn || b >= nbk) return 0.0;
    const double *omd = om + (k + 1) * L.nx;
    if (a < L.nx) {
        double co = (b < L.nx) ? c.Ad()[a * L.nx + b] : c.Bd()[a * L.nu + (b - L.nx)];
        return -omd[a] * co;
    }
    if (b < L.nx) return 0.0;
    int ja = a - L.nx, jb = b - L.nx;
    double v = -cc * c.QDu()[jb * L.nu + ja];                // mirror of the upper block -QDu
    if (ja == 0 && jb == L.nu - 1) v -= om[L.rdu + L.nu + k * L.nu + L.nu - 1];
    return v;
}

// ------------------------------------------------------------------------------------------------
// Block LDL' of the block-tridiagonal K (stage blocks NB x NB, NB = 16 or 32):
//      S_0 = K_00,   Mh_k = K_{k,k-1} S_{k-1}^-1,   S_k = K_kk - Mh_k K_{k,k-1}'
// Solve K x = b:    yh_0 = b_0,  yh_k = b_k - Mh_k yh_{k-1};   w_k = S_k^-1 yh_k;
//                   x_{N-1} = w_{N-1},  x_k = w_k - Mh_{k+1}' x_{k+1}.
// The factor is stored in the operand order of the matrix-core instruction the sweeps use, v_mfma_f64_4x4x4_4b_f64
// (four independent 4x4x4 products per instruction; 31 cycles dependent latency measured, against 83 for the
// 16x16x4 shape, and a quarter of its pipe time).  Layouts probed on gfx950 (scripts/probe_mfma4.hip):
//     A[blk][i][k] in lane 16k + 4blk + i,   B[blk][k][j] in lane 16k + 4blk + j,   D[blk][i][j] in lane 16i + 4blk + j.
// A 16x16 block M times a 16-vector v, as 4x4 sub-blocks M_IJ: step s = 0..3 computes, in block slot b,
// M_{b,(b+s)%4} * v_{(b+s)%4} (B operand = the sub-vector replicated over j) and accumulates y_b = sum_J M_bJ v_J.
// The result y[4b+i] sits in lane 16i + 4b + j, which is exactly where step 0 of the NEXT product wants its B operand
// (lane 16k + 4b + j holds v[4b+k]); steps 1..3 need the sub-vector of the neighbouring block slot, a rotation of each
// 16-lane row by 4, 8, 12 lanes: DPP row_ror.  So a stage vector is ONE double per lane, stage outputs feed the next
// stage through three DPP rotations and no LDS traffic, and a lane's four fragment values (one per step) are
// contiguous: fragment element (r, c) -> lane 16(c&3) + 4(r>>2) + (r&3), step ((c>>2) - (r>>2)) & 3.
// Per stage k:  [ forward matrix | S_k^-1 ]   (2 NB^2 doubles, 32 B per lane per block); the back substitution
// applies the forward matrix of the neighbouring stage TRANSPOSED from the same fragments (frag_matvec_T).
// ------------------------------------------------------------------------------------------------
// Optimisation barriers: values the compiler would otherwise hoist out of the ADMM iteration loop (loop-invariant
// loads and address arithmetic of the sweeps) and keep live across ALL phases, pushing the kernel into scratch spills.
template <class T> __device__ __forceinline__ T *opaque_ptr(T *p) {      // workgroup-uniform pointer, pinned to scalar registers
    unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    p = (T *)(((unsigned long long)hi << 32) | lo);
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ int opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdouble;     // explicit global address space: plain global_load/store,
typedef __attribute__((address_space(1))) const double cgdouble;  // not flat_* (which also counts on lgkmcnt)
typedef __attribute__((address_space(1))) const d4 cgd4;

template <int NB>
__device__ __forceinline__ int frag_pos(int r, int cidx) {
    constexpr int NBLK = NB / 16;
    const int bi = r >> 4, bj = cidx >> 4, rr = r & 15, cc = cidx & 15;
    const int b = rr >> 2, i = rr & 3, J = cc >> 2, k = cc & 3;
    const int sft = (J - b) & 3;                       // MFMA step in which 4x4 block (b, J) is used
    const int lane = k * 16 + b * 4 + i;
    return (bi * NBLK + bj) * 256 + lane * 4 + sft;
}

// TWISTED (two-sided) elimination: stages 0..mid-1 are eliminated top-down, stages N-1..mid+1 bottom-up, the
// middle stage mid = N/2 last, so that two waves can sweep the two half-chains concurrently (half the
// sequential depth).  With Sn = S^-1 of the neighbour eliminated just before,
//   top    k < mid:  Mh_k = K_{k,k-1} Sn_{k-1},   S_k = K_kk - Mh_k K_{k,k-1}'
//   bottom k > mid:  Mt_k = K_{k,k+1} Sn_{k+1},   S_k = K_kk - Mt_k K_{k,k+1}'        (K_{k,k+1} = K_{k+1,k}')
//   middle        :  S_mid = K_mm - Mh_mid K_{mid,mid-1}' - Mt_mid K_{mid,mid+1}'
// Per-stage factor slots (fragments, see above):   slot 0: forward matrix   slot 1: S_k^-1
//   top:    slot0 = -Mh_k        bottom: slot0 = -Mt_k        middle: slot0 = -Mh_mid, and its second forward matrix
//   -Mt_mid in slot 0 of stage 0 (which has none of its own).  Back substitution: x_k += slot0(k+1)' x_{k+1} in the
//   top half, x_k += slot0(k-1)' x_{k-1} in the bottom half (-Mt_mid' for k = mid+1).
// W: LDS workspace of 6*NB*NB doubles.  Returns (uniformly) 0, or 1 if a pivot was not positive.
struct BorderPtrs { double *Bb, *Zb, *Sig, *red; };

template <int NB> __device__ void border_factor(const Ctx &, const double *, const double *, double, const double *, double *, double *, double *, double *, double *, double *);

template <int NB>
__device__ int factor_all(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag, BorderPtrs bp) {
    const Lay &L = c.L;
    double *S = W, *Ks = W + NB * NB, *Mh = W + 2 * NB * NB, *SnA = W + 3 * NB * NB, *Li = W + 4 * NB * NB, *SnB = W + 5 * NB * NB;
    const int tid = threadIdx.x;
    const int N = L.N, mid = N / 2;
    if (tid == 0) *iflag = 0;
    // S -= (Ks Sn) Ks' for the neighbour on side `up` (true: k-1, false: k+1); stores the two fragment copies
    auto eliminate_neighbour = [&](int k, bool up, const double *Sn) {
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {
            int a = e / NB, b = e % NB;
            Ks[e] = up ? kkt_sub_entry(c, om, cc, k - 1, a, b) : kkt_sub_entry(c, om, cc, k, b, a);
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {              // Mh = Ks * Sn
            int a = e / NB, b = e % NB;
            double acc = 0.0;
            for (int l = 0; l < NB; ++l) acc += Ks[a * NB + l] * Sn[l * NB + b];
            Mh[e] = acc;
        }
        __syncthreads();
        const int fwd_stage = (k == mid && !up) ? 0 : k;
        for (int e = tid; e < NB * NB; e += NT) {              // S -= Mh * Ks'
            int a = e / NB, b = e % NB;
            double acc = 0.0;
            for (int l = 0; l < NB; ++l) acc += Mh[a * NB + l] * Ks[b * NB + l];
            S[e] -= acc;
            // forward matrix of stage k (the backward sweep applies the same fragment transposed); the middle stage's
            // second forward matrix lives in the otherwise unused slot 0 of stage 0
            F[(size_t)fwd_stage * L.fstage + frag_pos<NB>(a, b)] = -Mh[e];
        }
        __syncthreads();
    };
    auto stage = [&](int k, bool use_up, bool use_down, double *SnOut) {
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {
            S[e] = kkt_diag_entry(c, om, sv, cc, k, e / NB, e % NB);
            Li[e] = 0.0;
            if (k == N - 1) F[(size_t)k * L.fstage + e] = 0.0;      // the last stage has no forward matrix (stage 0's slot holds the middle's second one)
        }
        if (use_up) eliminate_neighbour(k, true, SnA);
        if (use_down) eliminate_neighbour(k, false, SnB);
        __syncthreads();
        // Cholesky of S (lower), right-looking
        for (int j = 0; j < NB; ++j) {
            double d = S[j * NB + j];
            if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
            d = sqrt(d);
            __syncthreads();
            for (int i = j + tid; i < NB; i += NT) S[i * NB + j] = (i == j) ? d : S[i * NB + j] / d;
            __syncthreads();
            const int rem = NB - 1 - j;
            for (int e = tid; e < rem * rem; e += NT) {
                int i = j + 1 + e / rem, l = j + 1 + e % rem;
                if (l <= i) S[i * NB + l] -= S[i * NB + j] * S[l * NB + j];
            }
            __syncthreads();
        }
        if (tid < NB) {                                       // Li = L^-1, one thread per column
            const int col = tid;
            Li[col * NB + col] = 1.0 / S[col * NB + col];
            for (int i = col + 1; i < NB; ++i) {
                double acc = 0.0;
                for (int l = col; l < i; ++l) acc += S[i * NB + l] * Li[l * NB + col];
                Li[i * NB + col] = -acc / S[i * NB + i];
            }
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {              // S^-1 = Li' Li
            int a = e / NB, b = e % NB;
            double acc = 0.0;
            for (int l = max(a, b); l < NB; ++l) acc += Li[l * NB + a] * Li[l * NB + b];
            SnOut[e] = acc;
            F[(size_t)k * L.fstage + NB * NB + frag_pos<NB>(a, b)] = acc;
        }
    };
    for (int k = 0; k < mid; ++k) stage(k, k > 0, false, SnA);
    for (int k = N - 1; k > mid; --k) stage(k, false, k < N - 1, SnB);
    stage(mid, true, true, SnA);
    __syncthreads();
    if (L.border) border_factor<NB>(c, om, sv, cc, F, bp.Bb, bp.Zb, bp.Sig, W, W + L.m, bp.red);
    return *iflag;
}

// The sweeps work on Tc: the x,u part of the right-hand side / solution in STAGE-MAJOR PADDED layout,
// Tc[k*NB + a] = element a of stage k (a < nx: x_k[a]; nx <= a < nb: u_k[a-nx]; everything else is padding
// and stays exactly zero because the factor is the identity there).  In the operand layout lane 16k + 4b + j
// holds element 4b + k (+16 per block): one 8-byte LDS read per lane and block.
template <int NB>
__device__ __forceinline__ void vec_load(const double *tb, int k, double *v) {
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) v[bi] = tb[k * NB + bi * 16];
}
template <int NB>
__device__ __forceinline__ void vec_store(double *tb, int k, const double *v, bool writer) {
    if (writer) {
#pragma unroll
        for (int bi = 0; bi < NB / 16; ++bi) tb[k * NB + bi * 16] = v[bi];
    }
}
// per-lane base of a stage vector in Tc: lane 16k + 4b + j holds element 4b + k
__device__ __forceinline__ int vec_lane_offset(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
__device__ __forceinline__ bool vec_lane_writer(int lane) { return (lane & 3) == 0; }

// rotate every 16-lane row by 4*sft lanes: lane (k, b, j) receives the value of lane (k, (b+sft)%4, j)
template <int SFT>
__device__ __forceinline__ double rot_blocks(double x) {
    if (SFT == 0) return x;
    constexpr int CTRL = 0x120 | (16 - 4 * SFT);          // row_ror:n gives dst[i] = src[(i - n) mod 16]
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)xi, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(xi >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// out[bi] += sum_bj A(bi,bj) * in[bj]   with A given as fragments (one d4 per block per lane)
template <int NB>
__device__ __forceinline__ void frag_matvec(const d4 *A, const double *in, double *out) {
    constexpr int NBLK = NB / 16;
#pragma unroll
    for (int bj = 0; bj < NBLK; ++bj) {
        const double r0 = in[bj], r1 = rot_blocks<1>(in[bj]), r2 = rot_blocks<2>(in[bj]), r3 = rot_blocks<3>(in[bj]);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) {
            const d4 a = A[bi * NBLK + bj];
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], r0, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, out[bi], 0, 0, 0);
        }
    }
}

template <int NB>
__device__ __forceinline__ void frag_load(const double *Fm, int lane, d4 *A) {
    constexpr int NBLK = NB / 16;
#pragma unroll
    for (int b = 0; b < NBLK * NBLK; ++b) A[b] = *(cgd4 *)(Fm + b * 256 + lane * 4);
}
#ifdef MPCQP_ABL_NOSINVLOAD
template <int NB>
__device__ __forceinline__ void frag_load_sinv(const double *Fm, int lane, d4 *A) {
#pragma unroll
    for (int b = 0; b < (NB / 16) * (NB / 16); ++b) A[b] = d4{1e-3 * lane, 1e-3, 2e-3, 3e-3};
}
#else
#define frag_load_sinv frag_load
#endif

// The TRANSPOSED product from the same fragments:  out[bj] += sum_bi A(bi,bj)' * in[bi].
// The backward substitution needs Mh' where the forward elimination needed Mh; the MFMA always contracts over the
// index that sits in the 16-lane-row position of the operand layout (the column of the stored block), so the
// transposed product is done on the vector ALU instead -- and the factor stream loses its third block per stage:
//   xl        lane (k,b,j) <- element 4b+j of `in`            (one cross-lane permute of the stage vector)
//   p_s = a[s] * xl                                            = M[4b+j][4((b+s)&3)+k] * in[4b+j]
//   t   = p_0 + rot_3(p_1) + rot_2(p_2) + rot_1(p_3)           block (b-s, b) contributes to output block b
//   out += sum over the four lanes j of t                      (two DPP quad steps), again replicated over j
__device__ __forceinline__ double lane_permute(double x, int byte_addr) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)xi), hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(xi >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)xi, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(xi >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int transpose_lane_addr(int lane) { return 4 * (16 * (lane & 3) + (lane & 12) + (lane >> 4)); }
template <int NB>
__device__ __forceinline__ void frag_matvec_T(const d4 *A, const double *in, double *out, int perm_addr) {
    constexpr int NBLK = NB / 16;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) {
        const double xl = lane_permute(in[bi], perm_addr);
#pragma unroll
        for (int bj = 0; bj < NBLK; ++bj) {
            const d4 a = A[bi * NBLK + bj];
            double t = (a[0] * xl + rot_blocks<3>(a[1] * xl)) + (rot_blocks<2>(a[2] * xl) + rot_blocks<1>(a[3] * xl));
            t += dpp_move<0xB1>(t);                            // quad_perm [1,0,3,2]
            t += dpp_move<0x4E>(t);                            // quad_perm [2,3,0,1]
            out[bj] += t;
        }
    }
}

template <int NB> struct SweepCfg {
    static constexpr int NF = (NB / 16) * (NB / 16);
#ifndef MPCQP_DEPTH
#define MPCQP_DEPTH 4
#endif
    static constexpr int DEPTH = MPCQP_DEPTH;                  // factor stages kept in flight in registers (even)
};

// The sweeping waves are dependent MFMA chains: two of them on one SIMD share its matrix pipe and slow each other
// down.  Workgroups that are co-resident on a CU (dispatch order: block b -> XCD b%8, CU (b/8)%32) therefore rotate
// which of their waves does what, so that the sweepers of the four co-resident workgroups spread over the four
// SIMDs.  Purely a speed matter: any placement gives the same results.
__device__ __forceinline__ int logical_wave() {
#ifdef MPCQP_NO_WAVE_ROTATION
    return threadIdx.x >> 6;
#else
    return ((threadIdx.x >> 6) - (blockIdx.x >> 8)) & (NWAVES - 1);
#endif
}

// Sequential sweep over `nsteps` stages by ONE wave: for i = 1..nsteps, k = first + dir*i:
//     forward elimination (TRANSPOSED = false):  Tc[k] <- Tc[k] + Fwd(k)        * Tc[k - dir]
//     back substitution   (TRANSPOSED = true) :  Tc[k] <- Tc[k] + Fwd(k - dir)' * Tc[k - dir]
// (Fwd(k) = slot 0 of stage k holds the negated factor block; `first_stage` >= 0 names the stage whose slot replaces
// Fwd(first) -- the middle stage's second forward matrix.)  The factor fragments of the next DEPTH stages are prefetched into a
// register ring; the running vector ping-pongs between two register sets (no copies between MFMAs).
template <int NB, bool TRANSPOSED>
__device__ __forceinline__ void chain_sweep(const int first, const int dir, const int nsteps,
                                            const int fstage, const double *F, const int first_stage, double *Tc) {
    constexpr int NBLK = NB / 16, NF = SweepCfg<NB>::NF, DEPTH = SweepCfg<NB>::DEPTH;
    const int lane = opaque_lane(threadIdx.x & 63);
    double *tb = Tc + vec_lane_offset(lane);
    const bool writer = vec_lane_writer(lane);
    const int perm_addr = transpose_lane_addr(lane);
    auto stage_of = [&](int i) { return first + dir * i; };
    auto frag_of = [&](int i) {                                // (offsets, not pointer selects)
        int st = TRANSPOSED ? stage_of(i - 1) : stage_of(i);
        if (TRANSPOSED && i == 1 && first_stage >= 0) st = first_stage;
        return F + (size_t)st * fstage;
    };
    // The group loop below is branch-free on purpose: with conditionals around the refills the compiler can no longer
    // count the loads in flight across the back edge and falls back to s_waitcnt vmcnt(0) -- the whole memory latency
    // once per group.  Refills past the end re-read the last stage (clamped index), the tail group runs separately.
    auto frag_clamped = [&](int i) { return frag_of(i < nsteps ? i : nsteps); };
    d4 ring[DEPTH][NF];
    if (nsteps < 1) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) frag_load<NB>(frag_clamped(1 + d), lane, ring[d]);
    double va[NBLK], vb[NBLK];
    vec_load<NB>(tb, first, va);
    auto stage_step = [&](int i, int d) {
        const int k = stage_of(i);
        double *src = (d & 1) ? vb : va, *dst = (d & 1) ? va : vb;
        vec_load<NB>(tb, k, dst);
        if (TRANSPOSED) frag_matvec_T<NB>(ring[d], src, dst, perm_addr);
        else frag_matvec<NB>(ring[d], src, dst);
        vec_store<NB>(tb, k, dst, writer);
    };
    int i0 = 1;
    for (; i0 + DEPTH - 1 <= nsteps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            stage_step(i0 + d, d);
            frag_load<NB>(frag_clamped(i0 + d + DEPTH), lane, ring[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (i0 + d <= nsteps) stage_step(i0 + d, d);
}

// w_k = S_k^-1 yh_k for all stages (independent MFMA groups, dealt to the four waves).  Wave 0 first finishes the
// forward elimination at the middle stage: yh_mid = b_mid - Mh_mid yh_{mid-1} - Mt_mid yh_{mid+1}; it owns the three
// stages around the middle (it reads yh_{mid-1}, yh_{mid+1}, so no other wave may overwrite them with w meanwhile).
// The other N-3 stages are spread so that the four waves finish together: wave 0 takes every 7th of them on top of
// its 3.5 stage-equivalents, waves 1..3 the rest in turn.  The wave's t-th stage in closed form:
static_assert(NWAVES == 4, "stage-to-wave map below assumes four waves");
__device__ __forceinline__ int sinv_stage(int wv, int t, int N, int mid) {        // -1: the wave has no t-th stage
    int j;                                                   // index among the stages outside {mid-1, mid, mid+1}
    if (wv == 0) j = 7 * t; else { const int p = 3 * t + (wv - 1); j = p + p / 6 + 1; }
    if (j >= N - 3) return -1;
    return j < mid - 1 ? j : j + 3;
}
// The fragment loads are software-pipelined two stages ahead; the first pair (and wave 0's five fragments around the
// middle) is requested BEFORE the barrier that ends the forward elimination (sinv_prefetch), so that waves 2 and 3,
// idle during the sweeps, have their data long before they may start.
template <int NB> struct SinvPre {
    d4 P0[SweepCfg<NB>::NF], P1[SweepCfg<NB>::NF];          // the wave's first two stages
    d4 A0[SweepCfg<NB>::NF], A2[SweepCfg<NB>::NF], Am[SweepCfg<NB>::NF], B0[SweepCfg<NB>::NF], B1[SweepCfg<NB>::NF];   // wave 0
    int nt, klast;                                           // number of stages of this wave, the last one (clamp target)
};
template <int NB>
__device__ __forceinline__ void sinv_prefetch(const int N, const int mid, const int fstage, const double *F, SinvPre<NB> &pre) {
    const int lane = opaque_lane(threadIdx.x & 63), wv = logical_wave();
    const double *Fs = F + NB * NB;
    int nt = 0, klast = 0;
    for (int t = 0; t < N; ++t) { const int k = sinv_stage(wv, t, N, mid); if (k < 0) break; klast = k; ++nt; }
    pre.nt = nt; pre.klast = klast;
    auto kc = [&](int t) { const int k = sinv_stage(wv, t, N, mid); return k < 0 ? klast : k; };
    if (wv == 0) {
        frag_load<NB>(F + (size_t)mid * fstage, lane, pre.A0);
        frag_load<NB>(F, lane, pre.A2);                      // the middle's second forward matrix (kept in stage 0's slot)
        frag_load_sinv<NB>(Fs + (size_t)mid * fstage, lane, pre.Am);
        frag_load_sinv<NB>(Fs + (size_t)(mid - 1) * fstage, lane, pre.B0);
        frag_load_sinv<NB>(Fs + (size_t)(mid + 1) * fstage, lane, pre.B1);
    }
    frag_load_sinv<NB>(Fs + (size_t)kc(0) * fstage, lane, pre.P0);
    frag_load_sinv<NB>(Fs + (size_t)kc(1) * fstage, lane, pre.P1);
}
template <int NB>
__device__ __forceinline__ void sinv_apply(const int N, const int mid, const int fstage, const double *F, double *Tc, SinvPre<NB> &pre) {
    constexpr int NBLK = NB / 16, NF = SweepCfg<NB>::NF;
    const int lane = opaque_lane(threadIdx.x & 63), wv = logical_wave();
    double *tb = Tc + vec_lane_offset(lane);
    const bool writer = vec_lane_writer(lane);
    auto apply = [&](int k, const d4 *A, bool valid) {       // (invalid: a clamped repeat of the last stage -- computed, not stored)
        double in[NBLK], out[NBLK];
        vec_load<NB>(tb, k, in);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) out[b] = 0.0;
        frag_matvec<NB>(A, in, out);
        vec_store<NB>(tb, k, out, writer && valid);
    };
    const double *Fs = F + NB * NB;
    const int nt = pre.nt, klast = pre.klast;
    auto kc = [&](int t) { const int k = sinv_stage(wv, t, N, mid); return k < 0 ? klast : k; };
    if (wv == 0) {
        double up[NBLK], dn[NBLK], acc[NBLK];
        vec_load<NB>(tb, mid, acc);
        vec_load<NB>(tb, mid - 1, up);
        vec_load<NB>(tb, mid + 1, dn);
        frag_matvec<NB>(pre.A0, up, acc);
        frag_matvec<NB>(pre.A2, dn, acc);
        vec_store<NB>(tb, mid, acc, writer);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        apply(mid, pre.Am, true); apply(mid - 1, pre.B0, true); apply(mid + 1, pre.B1, true);
    }
    d4 Q0[NF], Q1[NF];
    for (int t = 0; t < nt; t += 4) {                        // branch-free body (exact vmcnt waits), see chain_sweep
        frag_load_sinv<NB>(Fs + (size_t)kc(t + 2) * fstage, lane, Q0);
        frag_load_sinv<NB>(Fs + (size_t)kc(t + 3) * fstage, lane, Q1);
        apply(kc(t), pre.P0, true); apply(kc(t + 1), pre.P1, t + 1 < nt);
        frag_load_sinv<NB>(Fs + (size_t)kc(t + 4) * fstage, lane, pre.P0);
        frag_load_sinv<NB>(Fs + (size_t)kc(t + 5) * fstage, lane, pre.P1);
        apply(kc(t + 2), Q0, t + 2 < nt); apply(kc(t + 3), Q1, t + 3 < nt);
    }
}

// What the linear-system core needs to know about one instance.
struct CoreArgs { int N, fstage; const double *F; };
__device__ __forceinline__ CoreArgs core_args(const Lay &L, const double *F) {
    CoreArgs a; a.N = L.N; a.fstage = L.fstage; a.F = F; return a;
}

// Twisted solve: forward elimination of the two half-chains (waves 0, 1), S^-1 of every stage (all waves; wave 0
// closes the elimination at the middle first), back substitution outwards with the transposed forward matrices.
#ifdef MPCQP_RUN_TIMING
__device__ unsigned long long g_ticks[16];
#define TICK(i) { unsigned long long t_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&g_ticks[i], t_ - ttick); ttick = t_; }
#define TICK_START unsigned long long ttick = wall_clock64();
#else
#define TICK(i)
#define TICK_START
#endif

template <int NB>
__device__ __forceinline__ void kkt_core_sweeps(const CoreArgs &a, double *Tc) {
    const int N = a.N, fstage = a.fstage, mid = N / 2, wv = logical_wave();
    const double *F = a.F;
    TICK_START
    if (wv == 0) chain_sweep<NB, false>(0, +1, mid - 1, fstage, F, -1, Tc);               // stages 1 .. mid-1
    else if (wv == 1) chain_sweep<NB, false>(N - 1, -1, N - 2 - mid, fstage, F, -1, Tc);  // stages N-2 .. mid+1
    SinvPre<NB> pre;
    sinv_prefetch<NB>(N, mid, fstage, F, pre);
    __syncthreads();
    TICK(1)
    sinv_apply<NB>(N, mid, fstage, F, Tc, pre);
    __syncthreads();
    TICK(2)
    if (wv == 0) chain_sweep<NB, true>(mid, -1, mid, fstage, F, -1, Tc);                  // stages mid-1 .. 0
    else if (wv == 1) chain_sweep<NB, true>(mid, +1, N - 1 - mid, fstage, F, 0, Tc);           // stages mid+1 .. N-1
    __syncthreads();
    TICK(3)
}

// Tc <- K_xu^-1 Tc (eps already eliminated).  All threads call; barriers inside.  Waves 0 and 1 sweep the two
// half-chains of the twisted factorization concurrently.  Tc must be seen by the compiler as an LDS pointer
// (a pointer laundered through an integer becomes FLAT: flat LDS accesses count on vmcnt AND lgkmcnt and force a
// full s_waitcnt vmcnt(0) -- draining the factor prefetch -- before every stage).
template <int NB>
__device__ __forceinline__ void kkt_core(const CoreArgs &a, double *Tc) {
#ifndef MPCQP_ABL_NOCHAIN
    kkt_core_sweeps<NB>(a, Tc);
#endif
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Control horizon Nc < Np (mpc.py:513-517,540-543): the last input ub = u_{Nc-1} is held to the end of the
// horizon, so it couples to every later stage and K is block tridiagonal plus a border:
//     K = [ T  B ; B' C ],   T = K without ub (stage Nc-1 keeps only x),   B = K[:, ub],   C = K[ub, ub].
// With Z = T^-1 B and Sigma = C - B'Z (computed at factor time):  ub = Sigma^-1 (r2 - Z' r1),  y = T^-1 (r1 - B ub).
// B and C are taken entry by entry from the matrix-free operators (K = cP + diag(s) + A' diag(omega) A).
// ------------------------------------------------------------------------------------------------
__device__ double kkt_entry_generic(const Ctx &c, const double *om, const double *sv, double cc, int v, int w) {
    double acc = 0.0;
    P_row(c, v, [&](double co, int idx) { if (idx == w) acc += cc * co; });
    if (v == w) acc += sv[v];
    AT_row(c, v, [&](double cot, int r) {
        double arw = 0.0;
        A_row(c, r, [&](double co, int idx) { if (idx == w) arw += co; });
        acc += cot * om[r] * arw;
    });
    return acc;
}

// flat variable index of padded slot (k, a), or -1 for padding / the border input
__device__ __forceinline__ int padded_var(const Lay &L, int k, int a) {
    if (a < L.nx) return k * L.nx + a;
    if (a < L.nb && k < L.NcT) return L.ou + k * L.nu + (a - L.nx);
    return -1;
}

template <int NB>
__device__ void border_factor(const Ctx &c, const double *om, const double *sv, double cc, const double *F,
                              double *Bb, double *Zb, double *Sig, double *W, double *Tc, double *red) {
    const Lay &L = c.L;
    const int tid = threadIdx.x, nu = L.nu, NP = L.N * NB;
    const int ub0 = L.ou + (L.Nc - 1) * L.nu;
    for (int idx = tid; idx < NP; idx += NT) {
        const int v = padded_var(L, idx / NB, idx % NB);
        for (int j = 0; j < nu; ++j) Bb[(size_t)j * NP + idx] = (v >= 0) ? kkt_entry_generic(c, om, sv, cc, v, ub0 + j) : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nu; ++j) {                         // Z_j = T^-1 B_j
        for (int idx = tid; idx < NP; idx += NT) Tc[idx] = Bb[(size_t)j * NP + idx];
        __syncthreads();
        kkt_core<NB>(core_args(L, F), Tc);
        for (int idx = tid; idx < NP; idx += NT) Zb[(size_t)j * NP + idx] = Tc[idx];
        __syncthreads();
    }
    // Sigma = C - B'Z, inverted by Gauss-Jordan (SPD, nu x nu) by one thread
    double *Sg = W;                                        // nu*nu doubles (W, the row work vector, is free here)
    for (int e = 0; e < nu * nu; ++e) {
        const int i = e / nu, j = e % nu;
        double vsum[1] = {0.0}, vmax[1] = {0.0};
        for (int idx = tid; idx < NP; idx += NT) vsum[0] += Bb[(size_t)i * NP + idx] * Zb[(size_t)j * NP + idx];
        block_reduce<1, 1>(vmax, vsum, red);
        if (tid == 0) Sg[e] = kkt_entry_generic(c, om, sv, cc, ub0 + i, ub0 + j) - vsum[0];
        __syncthreads();
    }
    if (tid == 0) {
        double *Iv = W + nu * nu;                          // scratch for the inverse
        for (int e = 0; e < nu * nu; ++e) Iv[e] = (e / nu == e % nu) ? 1.0 : 0.0;
        for (int p = 0; p < nu; ++p) {
            double d = 1.0 / Sg[p * nu + p];
            for (int j = 0; j < nu; ++j) { Sg[p * nu + j] *= d; Iv[p * nu + j] *= d; }
            for (int i = 0; i < nu; ++i) if (i != p) {
                double f = Sg[i * nu + p];
                for (int j = 0; j < nu; ++j) { Sg[i * nu + j] -= f * Sg[p * nu + j]; Iv[i * nu + j] -= f * Iv[p * nu + j]; }
            }
        }
        for (int e = 0; e < nu * nu; ++e) Sig[e] = Iv[e];
    }
    __syncthreads();
}

// Before the tridiagonal solve: Tc holds r1 in the padded slots and r2 in the (otherwise padding) u slots of stage
// Nc-1.  Computes ub, leaves it in ubar[] (LDS, nu doubles) and replaces r1 by r1 - B ub.
template <int NB>
__device__ void border_pre(const Lay &L, const double *Bb, const double *Zb, const double *Sig, double *Tc, double *ubar, double *red) {
    const int tid = threadIdx.x, nu = L.nu, NP = L.N * NB;
    const int slot = (L.Nc - 1) * NB + L.nx;
    for (int j = 0; j < nu; ++j) {
        double vsum[1] = {0.0}, vmax[1] = {0.0};
        for (int idx = tid; idx < NP; idx += NT) vsum[0] += Zb[(size_t)j * NP + idx] * Tc[idx];     // Z is zero in the r2 slots
        block_reduce<1, 1>(vmax, vsum, red);
        if (tid == 0) ubar[nu + j] = Tc[slot + j] - vsum[0];
        __syncthreads();
    }
    if (tid < nu) { double a = 0.0; for (int j = 0; j < nu; ++j) a += Sig[tid * nu + j] * ubar[nu + j]; ubar[tid] = a; }
    __syncthreads();
    for (int idx = tid; idx < NP; idx += NT) {
        double a = Tc[idx];
        for (int j = 0; j < nu; ++j) a -= Bb[(size_t)j * NP + idx] * ubar[j];
        Tc[idx] = a;
    }
    if (tid < nu) Tc[slot + tid] = 0.0;
    __syncthreads();
}
__device__ __forceinline__ void border_post(const Lay &L, int NB, double *Tc, const double *ubar) {
    if ((int)threadIdx.x < L.nu) Tc[(L.Nc - 1) * NB + L.nx + threadIdx.x] = ubar[threadIdx.x];
    __syncthreads();
}

// Generic front end (verification kernel): flat rhs (global) -> flat solution `out` (global, n doubles).
// Tc: LDS, N*NB doubles.
template <int NB>
__device__ void kkt_solve(const Ctx &c, const double *om, const double *sv, double cc, const double *F,
                          const double *rg, double *Tc, double *out, BorderPtrs bp, double *ubar) {
    const Lay &L = c.L;
    const double cef = cc * c.eps_feas();
    for (int idx = threadIdx.x; idx < L.N * NB; idx += NT) {
        int k = idx / NB, a = idx % NB;
        double v = 0.0;
        if (a < L.nx) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e];
            double te = rg[L.oe + e] / (cef + sv[L.oe + e] + ws);
            out[L.oe + e] = te;
            v = rg[e] - ws * te;
        } else if (a < L.nb && k < L.Nc) v = rg[L.ou + k * L.nu + (a - L.nx)];
        Tc[idx] = v;
    }
    __syncthreads();
    if (L.border) border_pre<NB>(L, bp.Bb, bp.Zb, bp.Sig, Tc, ubar, bp.red);
    kkt_core<NB>(core_args(L, F), Tc);
    if (L.border) border_post(L, NB, Tc, ubar);
    for (int idx = threadIdx.x; idx < L.N * NB; idx += NT) {
        int k = idx / NB, a = idx % NB;
        if (a < L.nx) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e];
            double xe = Tc[idx];
            out[e] = xe;
            out[L.oe + e] -= (ws / (cef + sv[L.oe + e] + ws)) * xe;
        } else if (a < L.nb && k < L.Nc) out[L.ou + k * L.nu + (a - L.nx)] = Tc[idx];
    }
    __syncthreads();
}

// rho vector -> metric.  Constraint types are decided on the SCALED bounds, as OSQP does.
__device__ __forceinline__ int row_type(double E, double lo, double hi) {
    double ls = E * lo, us = E * hi;
    if (ls < -QP_INFTY * MIN_SCALING && us > QP_INFTY * MIN_SCALING) return -1;
    if (us - ls < RHO_TOL) return 1;
    return 0;
}
__device__ __forceinline__ double row_rho(int type, double rho) { return type < 0 ? RHO_MIN : (type > 0 ? RHO_EQ_OVER_RHO_INEQ * rho : rho); }

__device__ __forceinline__ BorderPtrs border_ptrs(const Lay &L, const Ptrs &P, double *red) {
    BorderPtrs bp; bp.red = red;
    const size_t npb = (size_t)L.nu * L.N * L.NB;
    bp.Bb = L.border ? P.Bb + blockIdx.x * npb : nullptr;
    bp.Zb = L.border ? P.Zb + blockIdx.x * npb : nullptr;
    bp.Sig = L.border ? P.Sig + (size_t)blockIdx.x * L.nu * L.nu : nullptr;
    return bp;
}

// Shared prologue: stage the hot model prefix and the step data in LDS.
struct Smem {
    double *T, *Qv, *hot, *x0s, *um1s, *red, *tv;
    int *iflag;
};
__device__ __forceinline__ double *carve(double *&p, int n) { double *r = p; p += n; return r; }
template <class PT>
__device__ void smem_common(const Lay &L, const PT &P, double *&p, Smem &S) {
    S.T = carve(p, L.tsz);
    S.Qv = (double *)P.qv + (size_t)blockIdx.x * (L.n_x + L.n_u);
    S.hot = carve(p, L.hot_sz);
    S.x0s = carve(p, L.nx);
    S.um1s = carve(p, L.nu);
    S.red = carve(p, 64);
    S.tv = carve(p, 64);
    S.iflag = (int *)carve(p, 2);
}
__host__ __device__ inline int smem_common_doubles(const Lay &L) { return L.tsz + L.hot_sz + L.nx + L.nu + 64 + 64 + 2; }

__device__ void load_common(const Lay &L, const double *model, const double *step, Smem &S) {
    for (int i = threadIdx.x; i < L.hot_sz; i += NT) S.hot[i] = model[i];
    for (int i = threadIdx.x; i < L.nx; i += NT) S.x0s[i] = step[i];
    for (int i = threadIdx.x; i < L.nu; i += NT) S.um1s[i] = step[L.nx + i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// setup kernel: Ruiz equilibration (OSQP, 10 passes), rho vector, metric, first factorization.
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(NT) void k_setup(Lay L, Ptrs P, mpcqp_settings S_) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model};
    build_q(c, step, S.Qv);
    double *D = P.D + (size_t)b * L.n, *E = P.E + (size_t)b * L.m, *Dt = P.Dt + (size_t)b * L.n, *Et = P.Et + (size_t)b * L.m;
    for (int j = tid; j < L.n; j += NT) D[j] = 1.0;
    for (int r = tid; r < L.m; r += NT) E[r] = 1.0;
    double cc = 1.0;
    __syncthreads();
    for (int it = 0; it < S_.scaling; ++it) {
        for (int j = tid; j < L.n; j += NT) {
            double pn = 0.0, an = 0.0;
            P_row(c, j, [&](double co, int idx) { pn = fmax(pn, fabs(co) * D[idx]); });
            AT_row(c, j, [&](double co, int row) { an = fmax(an, fabs(co) * E[row]); });
            pn *= cc * D[j]; an *= D[j];
            Dt[j] = 1.0 / sqrt(limit_scaling(fmax(pn, an)));
        }
        for (int r = tid; r < L.m; r += NT) {
            double en = 0.0;
            A_row(c, r, [&](double co, int idx) { en = fmax(en, fabs(co) * D[idx]); });
            Et[r] = 1.0 / sqrt(limit_scaling(en * E[r]));
        }
        __syncthreads();
        for (int j = tid; j < L.n; j += NT) D[j] *= Dt[j];
        for (int r = tid; r < L.m; r += NT) E[r] *= Et[r];
        __syncthreads();
        double vmax[1] = {0.0}, vsum[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double pn = 0.0;
            P_row(c, j, [&](double co, int idx) { pn = fmax(pn, fabs(co) * D[idx]); });
            vsum[0] += cc * D[j] * pn;
            double qj = (j < L.oe) ? S.Qv[j] : 0.0;
            vmax[0] = fmax(vmax[0], fabs(cc * D[j] * qj));
        }
        block_reduce<1, 1>(vmax, vsum, S.red);
        double ct = vsum[0] / (double)L.n;
        double qn = limit_scaling(vmax[0]);
        ct = limit_scaling(fmax(ct, qn));
        cc *= 1.0 / ct;
    }
    // rho vector / metric
    double rho = S_.rho;
    double *om = P.omega + (size_t)b * L.m, *sv = P.s + (size_t)b * L.n;
    int *ct = P.ctype + (size_t)b * L.m;
    for (int r = tid; r < L.m; r += NT) {
        double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
        int t = row_type(E[r], lo, hi);
        ct[r] = t;
        om[r] = row_rho(t, rho) * E[r] * E[r];
    }
    for (int j = tid; j < L.n; j += NT) sv[j] = S_.sigma / (D[j] * D[j]);
    if (tid == 0) { P.c[b] = cc; P.rho[b] = rho; }
    __syncthreads();
    int bad = factor_all<NB>(c, om, sv, cc, P.F + (size_t)b * P.fsz, S.T, S.iflag, border_ptrs(L, P, S.red));
    // cold start
    for (int j = tid; j < L.n; j += NT) { P.x[(size_t)b * L.n + j] = 0.0; P.xo[(size_t)b * L.n + j] = 0.0; }
    for (int r = tid; r < L.m; r += NT) { P.z[(size_t)b * L.m + r] = 0.0; P.y[(size_t)b * L.m + r] = 0.0; P.yo[(size_t)b * L.m + r] = 0.0; }
    if (tid == 0) {
        mpcqp_info inf; inf.status = bad ? MPCQP_NON_CVX : MPCQP_UNSOLVED; inf.iter = 0; inf.rho_updates = 0; inf.reserved = 0;
        inf.obj_val = 0; inf.pri_res = 0; inf.dua_res = 0; inf.rho = rho;
        P.info[b] = inf;
    }
}

// ------------------------------------------------------------------------------------------------
// A solve has three phases (bodies below; k_mpc_run strings them together per instance):
//   begin  once per solve : q refresh from (x0, u_{-1}, xref), constraint types, per-solve bookkeeping
//   admm   per round      : check_termination ADMM iterations -- the hot loop, nothing else in it
//   check  per round      : residuals, termination, infeasibility certificates, rho adaptation + refactor
// ------------------------------------------------------------------------------------------------
enum { COLD_CHECK = 1, COLD_RHO = 2, COLD_FINAL = 4, COLD_PLAIN = 8 };

// Refactorization from inside a solve: a non-inlined function with a register allocation of its own (defined with the
// other phases of k_mpc_run below), so that this rare, register-hungry path does not push the residual evaluation
// and the per-solve prologue into scratch spills.
template <int NB> __device__ void run_factor_phase();

template <int NB>
__device__ __forceinline__ void begin_body(const Lay &L, const Ptrs &P, const mpcqp_settings &S_, Smem &S, int plain) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    Ctx c{L, S.hot, model};
    build_q(c, step, S.Qv);
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    if (!(S_.warm_start || plain)) {
        for (int j = tid; j < L.n; j += NT) gx[j] = 0.0;
        for (int r = tid; r < L.m; r += NT) { gz[r] = 0.0; gy[r] = 0.0; }
    }
    // constraint types (bounds may have changed since the last factorization)
    double *om = P.omega + (size_t)b * L.m;
    const double *sv = P.s + (size_t)b * L.n, *E = P.E + (size_t)b * L.m;
    int *ctp = P.ctype + (size_t)b * L.m;
    const double rho = P.rho[b];
    int changed = 0;
    for (int r = tid; r < L.m; r += NT) {
        double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
        int t = row_type(E[r], lo, hi);
        if (t != ctp[r]) { changed = 1; ctp[r] = t; om[r] = row_rho(t, rho) * E[r] * E[r]; }
    }
    changed = __syncthreads_or(changed);
    if (changed) run_factor_phase<NB>();
    if (tid == 0) {
        mpcqp_info inf; inf.status = MPCQP_UNSOLVED; inf.iter = 0; inf.rho_updates = 0; inf.reserved = 0;
        inf.obj_val = 0.0; inf.pri_res = 0.0; inf.dua_res = 0.0; inf.rho = rho;
        P.info[b] = inf;
    }
}

// Returns 1 (to every thread) if the instance has terminated.
template <int NB>
__device__ __forceinline__ int check_body(const Lay &L, const Ptrs &P, const mpcqp_settings &S_, Smem &S, int iter, int mode,
                                          const double *Xl, const double *Zl, const double *Yl) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz;
    Ctx c{L, S.hot, model};
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    // the iterate: the LDS copy the last ADMM round left behind (small problems), else global memory
    const double *X = Xl ? Xl : gx, *Z = Zl ? Zl : gz, *Y = Yl ? Yl : gy;
    double *om = P.omega + (size_t)b * L.m;
    const double *sv = P.s + (size_t)b * L.n;
    const double *D = P.D + (size_t)b * L.n, *E = P.E + (size_t)b * L.m;
    const double *dxg = P.dx + (size_t)b * L.n, *dyg = P.dy + (size_t)b * L.m;
    const int *ctp = P.ctype + (size_t)b * L.m;
    const double cc = P.c[b];
    double rho = P.rho[b];
    int status = MPCQP_UNSOLVED;
    double obj_val, pri_res, dua_res;

    // ---- OSQP update_info: objective, unscaled residuals, and the scaled norms the rho estimate needs
    // vmax: 0 pri, 1 |Ax|, 2 |z|, 3 dua, 4 |Px|, 5 |A'y|, 6 |q|; scaled: 7 pri, 8 max(|EAx|,|Ez|), 9 dua, 10 max(|cD(..)|)
    double nrm[11], vsum[1] = {0.0};
#pragma unroll
    for (int i = 0; i < 11; ++i) nrm[i] = 0.0;
    for (int r = tid; r < L.m; r += NT) {
        double ax = 0.0;
        A_row(c, r, [&](double co, int idx) { ax += co * X[idx]; });
        double z = Z[r], d = ax - z, e = E[r];
        nrm[0] = fmax(nrm[0], fabs(d)); nrm[1] = fmax(nrm[1], fabs(ax)); nrm[2] = fmax(nrm[2], fabs(z));
        nrm[7] = fmax(nrm[7], fabs(e * d)); nrm[8] = fmax(nrm[8], fmax(fabs(e * ax), fabs(e * z)));
    }
    for (int j = tid; j < L.n; j += NT) {
        double px = 0.0, aty = 0.0;
        P_row(c, j, [&](double co, int idx) { px += co * X[idx]; });
        AT_row(c, j, [&](double co, int row) { aty += co * Y[row]; });
        double qj = (j < L.oe) ? S.Qv[j] : 0.0, xj = X[j];
        double d = px + qj + aty, cd = cc * D[j];
        nrm[3] = fmax(nrm[3], fabs(d)); nrm[4] = fmax(nrm[4], fabs(px)); nrm[5] = fmax(nrm[5], fabs(aty)); nrm[6] = fmax(nrm[6], fabs(qj));
        nrm[9] = fmax(nrm[9], fabs(cd * d));
        nrm[10] = fmax(nrm[10], fmax(fabs(cd * qj), fmax(fabs(cd * aty), fabs(cd * px))));
        vsum[0] += xj * (0.5 * px + qj);
    }
    block_reduce<11, 1>(nrm, vsum, S.red);
    obj_val = vsum[0]; pri_res = nrm[0]; dua_res = nrm[3];

    // ---- OSQP's infeasibility certificates (paper section 3.5) on the last increments, in unscaled terms
    auto primal_infeasible = [&](double eps) -> bool {
        // v = c * delta_y (= E * scaled delta_y), projected on the polar of the recession cone of [l,u]
        double vmax[1] = {0.0}, vs[1] = {0.0};
        for (int r = tid; r < L.m; r += NT) {
            double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
            double e = E[r], v = cc * dyg[r];
            if (e * hi > QP_INFTY * MIN_SCALING) { if (e * lo < -QP_INFTY * MIN_SCALING) v = 0.0; else v = fmin(v, 0.0); }
            else if (e * lo < -QP_INFTY * MIN_SCALING) v = fmax(v, 0.0);
            S.T[r] = v;
            vmax[0] = fmax(vmax[0], fabs(v));
            vs[0] += hi * fmax(v, 0.0) + lo * fmin(v, 0.0);
        }
        block_reduce<1, 1>(vmax, vs, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vs[0] < -eps * nd)) return false;
        double amax[1] = {0.0}, dummy[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double a = 0.0; AT_row(c, j, [&](double co, int row) { a += co * S.T[row]; });
            amax[0] = fmax(amax[0], fabs(a));
        }
        block_reduce<1, 1>(amax, dummy, S.red);
        return amax[0] < eps * nd;
    };
    auto dual_infeasible = [&](double eps) -> bool {
        double vmax[1] = {0.0}, vs[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double d = dxg[j];
            vmax[0] = fmax(vmax[0], fabs(d));
            vs[0] += ((j < L.oe) ? S.Qv[j] : 0.0) * d;
        }
        block_reduce<1, 1>(vmax, vs, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vs[0] < -eps * nd)) return false;
        double pmax[1] = {0.0}, bad[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double a = 0.0; P_row(c, j, [&](double co, int idx) { a += co * dxg[idx]; });
            pmax[0] = fmax(pmax[0], fabs(a));
        }
        for (int r = tid; r < L.m; r += NT) {
            double a = 0.0; A_row(c, r, [&](double co, int idx) { a += co * dxg[idx]; });
            double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
            double e = E[r];
            if ((e * hi < QP_INFTY * MIN_SCALING && a > eps * nd) || (e * lo > -QP_INFTY * MIN_SCALING && a < -eps * nd)) bad[0] = 1.0;
        }
        block_reduce<1, 1>(pmax, bad, S.red);
        return (pmax[0] < eps * nd) && (bad[0] == 0.0);
    };
    auto check_termination = [&](bool approx) -> bool {
        double ea = S_.eps_abs, er = S_.eps_rel, epi = S_.eps_prim_inf, edi = S_.eps_dual_inf;
        if (pri_res > QP_INFTY || dua_res > QP_INFTY) { status = MPCQP_NON_CVX; obj_val = NAN; return true; }
        if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
        bool pc = pri_res < ea + er * fmax(nrm[2], nrm[1]);
        bool dc = dua_res < ea + er * fmax(fmax(nrm[6], nrm[5]), nrm[4]);
        bool pic = false, dic = false;
        if (!pc) pic = primal_infeasible(epi);
        if (!dc) dic = dual_infeasible(edi);
        if (pc && dc) { status = approx ? MPCQP_SOLVED_INACCURATE : MPCQP_SOLVED; return true; }
        if (pic) { status = approx ? MPCQP_PRIMAL_INFEASIBLE_INACCURATE : MPCQP_PRIMAL_INFEASIBLE; obj_val = QP_INFTY; return true; }
        if (dic) { status = approx ? MPCQP_DUAL_INFEASIBLE_INACCURATE : MPCQP_DUAL_INFEASIBLE; obj_val = -QP_INFTY; return true; }
        return false;
    };

    int term = 0, rho_upd = 0;
    if (mode & COLD_PLAIN) { status = MPCQP_UNSOLVED; term = 1; }
    else {
        if (mode & COLD_CHECK) term = check_termination(false) ? 1 : 0;
        if (!term && (mode & COLD_FINAL)) {             // iteration limit: OSQP retries with 10x looser tolerances
            if (!check_termination(true)) status = MPCQP_MAX_ITER_REACHED;
            term = 1;
        }
        if (!term && (mode & COLD_RHO)) {
            double pri = nrm[7] / (nrm[8] + 1e-10), dua = nrm[9] / (nrm[10] + 1e-10);
            double rn = fmin(fmax(rho * sqrt(pri / (dua + 1e-10)), RHO_MIN), RHO_MAX);
            if (rn > rho * S_.adaptive_rho_tolerance || rn < rho / S_.adaptive_rho_tolerance) {
                rho = rn;
                for (int r = tid; r < L.m; r += NT) om[r] = row_rho(ctp[r], rho) * E[r] * E[r];
                __syncthreads();
                run_factor_phase<NB>();
                rho_upd = 1;
            }
        }
    }
    __syncthreads();
    if (term) {      // solution, and the iterate the next warm start begins from
        const bool has_sol = !(status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                               status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX);
        double *xo = P.xo + (size_t)b * L.n, *yo = P.yo + (size_t)b * L.m;
        for (int j = tid; j < L.n; j += NT) { double v = gx[j]; xo[j] = has_sol ? v : NAN; if (!has_sol) gx[j] = 0.0; }
        for (int r = tid; r < L.m; r += NT) { double v = gy[r]; yo[r] = has_sol ? v : NAN; if (!has_sol) { gy[r] = 0.0; gz[r] = 0.0; } }
    }
    if (tid == 0) {
        mpcqp_info inf = P.info[b];
        inf.status = status; inf.iter = iter; inf.rho_updates += rho_upd; inf.reserved += 1;
        inf.obj_val = obj_val; inf.pri_res = pri_res; inf.dua_res = dua_res; inf.rho = rho;
        P.rho[b] = rho;
        if (term) {
            atomicAdd(&P.stats[0], (unsigned long long)iter); atomicAdd(&P.stats[1], (unsigned long long)inf.reserved);
            atomicAdd(&P.stats[2], (unsigned long long)inf.rho_updates); atomicAdd(&P.stats[3], 1ULL);
            inf.reserved = 0;
        }
        P.info[b] = inf;
    }
    return term;
}

// ---- hot-loop pieces.  NXT/NUT: compile-time nx/nu (0 = take them from the layout at run time).
template <int NXT> __device__ __forceinline__ int hx(const Lay &L) { return NXT ? NXT : L.nx; }
template <int NUT> __device__ __forceinline__ int hu(const Lay &L) { return NUT ? NUT : L.nu; }
template <int NXT> __device__ __forceinline__ int divx(const Lay &L, int v) { return NXT ? v / NXT : idiv(v, L.rnx); }
template <int NUT> __device__ __forceinline__ int divu(const Lay &L, int v) { return NUT ? v / NUT : idiv(v, L.rnu); }

// Steps (1)-(2) of the ADMM iteration with the slack elimination fused in:
//   W = omega z - c y                       (rows, flat; left behind by hot_rows_w / the previous hot_update)
//   rhs = s x - c q + A' W                  (variables)
//   te = rhs_eps / kappa -> W[soft row]     Tc[k][a] = rhs_x - omega_soft te  |  rhs_u  |  0 (padding)
// Small problems (REGV; m <= 4 NT rows and N NB <= 2 NT padded variables, the same condition as the LDS-resident
// iterate): a thread always handles the same rows r = tid + NT j and the same padded variables idx = tid + NT j, and
// what it needs of the iteration-invariant vectors omega, s, q stays in its registers for the whole round -- the
// parallel phases then touch no global memory at all (ten dependent global-load latencies per iteration otherwise).
struct HotRegs {
    double om_r[4];                                  // omega of the thread's rows
    double sv_e[2], qv_e[2];                         // per padded variable: its s and its linear cost q
    double om_s[2], sv_s[2];                         // (x part only) omega of its soft row, s of its slack
};
template <int NB, int NXT, int NUT>
__device__ __forceinline__ void load_hot_regs(const Lay &L, cgdouble *om, cgdouble *sv, cgdouble *qv, HotRegs &h) {
    const int tid = threadIdx.x, nx = hx<NXT>(L), nu = hu<NUT>(L);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = tid + NT * j; h.om_r[j] = r < L.m ? om[r] : 1.0; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + NT * j, k = idx / NB, a = idx % NB;
        h.sv_e[j] = 0.0; h.qv_e[j] = 0.0; h.om_s[j] = 1.0; h.sv_s[j] = 0.0;
        if (idx < L.N * NB) {
            if (a < nx) { const int e = k * nx + a; h.sv_e[j] = sv[e]; h.qv_e[j] = qv[e]; h.om_s[j] = om[L.rs + e]; h.sv_s[j] = sv[L.oe + e]; }
            else if (a < nx + nu && k < L.Nc) { const int cu = k * nu + a - nx; h.sv_e[j] = sv[L.ou + cu]; h.qv_e[j] = qv[L.n_x + cu]; }
        }
    }
}

// W = omega z - c y for the first iteration of a round (afterwards hot_update leaves it behind: a thread owns its rows).
template <bool REGV>
__device__ __forceinline__ void hot_rows_w(const Lay &L, cgdouble *om, const HotRegs &h, double cc, const double *Z, const double *Y, double *W) {
    const int tid = opaque_lane(threadIdx.x);
    if (REGV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int r = tid + NT * j; if (r < L.m) W[r] = h.om_r[j] * Z[r] - cc * Y[r]; }
    } else {
        for (int r = tid; r < L.m; r += NT) W[r] = om[r] * Z[r] - cc * Y[r];
    }
    __syncthreads();
}

template <int NB, int NXT, int NUT, bool REGV>
__device__ __forceinline__ void hot_rhs(const Lay &L, const double *hot, cgdouble *om, cgdouble *sv, cgdouble *qv, const HotRegs &h, double cc,
                                        const double *X, const double *Z, const double *Y, double *W, double *Tc) {
    const int tid = opaque_lane(threadIdx.x);        // (keeps the per-thread index arithmetic out of LICM's reach: hoisted, it spills)
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    const double cef = cc * hot[L.oeps];
    auto element = [&](int idx, double sve, double qve, double ws, double svs, bool have) {
        const int k = idx / NB, a = idx % NB;
        double v = 0.0;
        if (a < nx) {
            const int e = k * nx + a;
            if (!have) { sve = sv[e]; qve = qv[e]; ws = om[L.rs + e]; svs = sv[L.oe + e]; }
            double rx = sve * X[e] - cc * qve - W[e];
            if (k < L.Np) {
                const double *w1 = W + (k + 1) * nx;
#pragma unroll
                for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) rx += Ad[r * nx + a] * w1[r];
                if (!NXT) for (int r = 0; r < nx; ++r) rx += Ad[r * nx + a] * w1[r];
            }
            const double wsoft = W[L.rs + e];
            const double te = (svs * X[L.oe + e] + wsoft) / (cef + svs + ws);
            W[L.rs + e] = te;                      // only this thread ever reads W[soft row e]
            v = rx + wsoft - ws * te;
        } else if (a < nx + nu && k < L.Nc) {
            const int jj = a - nx, cu = k * nu + jj;
            if (!have) { sve = sv[L.ou + cu]; qve = qv[L.n_x + cu]; }
            double ru = sve * X[L.ou + cu] - cc * qve + W[L.ri + cu] - W[L.rdu + nu + cu];
            if (k == 0) ru += W[L.rdu + jj];
            if (cu > 0) ru += W[L.rdu + nu + cu - 1];
            const int s_end = (k == L.Nc - 1) ? L.Np : k + 1;
            for (int s = k + 1; s <= s_end; ++s) {
                const double *w1 = W + s * nx;
#pragma unroll
                for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) ru += Bd[r * nu + jj] * w1[r];
                if (!NXT) for (int r = 0; r < nx; ++r) ru += Bd[r * nu + jj] * w1[r];
            }
            v = ru;
        }
        Tc[idx] = v;
    };
    if (REGV) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int idx = tid + NT * j; if (idx < L.N * NB) element(idx, h.sv_e[j], h.qv_e[j], h.om_s[j], h.sv_s[j], true); }
    } else {
        for (int idx = tid; idx < L.N * NB; idx += NT) element(idx, 0.0, 0.0, 0.0, 0.0, false);
    }
    __syncthreads();
}

// Steps (4)-(6): slack back-substitution, zt = A xt, relaxation, projection on [l,u], dual update, x update.
template <int NB, int NXT, int NUT, bool REGV>
__device__ __forceinline__ void hot_update(const Lay &L, const double *hot, const double *x0s, const double *um1s,
                                           cgdouble *om, cgdouble *sv, const HotRegs &h, double cc, double alpha,
                                           double *X, double *Z, double *Y, double *W, const double *Tc, bool keep_delta, gdouble *dxg, gdouble *dyg) {
    const int tid = opaque_lane(threadIdx.x);        // (keeps the per-thread index arithmetic out of LICM's reach: hoisted, it spills)
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double cef = cc * hot[L.oeps];
    // eps_t = te - (omega_soft / kappa) x_t ; x update for the x and eps variables
    auto x_update = [&](int e, int k, int i, double ws, double svs) {
        const double xt = Tc[k * NB + i];
        const double et = W[L.rs + e] - (ws / (cef + svs + ws)) * xt;
        W[L.rs + e] = et;
        const double xo = X[e], eo = X[L.oe + e];
        const double xn = alpha * xt + (1.0 - alpha) * xo, en = alpha * et + (1.0 - alpha) * eo;
        X[e] = xn; X[L.oe + e] = en;
        if (keep_delta) { dxg[e] = xn - xo; dxg[L.oe + e] = en - eo; }
    };
    auto u_update = [&](int cu, int k, int jj) {
        const double uo = X[L.ou + cu];
        const double un = alpha * Tc[k * NB + nx + jj] + (1.0 - alpha) * uo;
        X[L.ou + cu] = un;
        if (keep_delta) dxg[L.ou + cu] = un - uo;
    };
    if (REGV) {                                          // same padded-variable -> thread map as hot_rhs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + NT * j, k = idx / NB, a = idx % NB;
            if (idx < L.N * NB) {
                if (a < nx) x_update(k * nx + a, k, a, h.om_s[j], h.sv_s[j]);
                else if (a < nx + nu && k < L.Nc) u_update(k * nu + a - nx, k, a - nx);
            }
        }
    } else {
        for (int e = tid; e < L.n_x; e += NT) {
            const int k = divx<NXT>(L, e), i = e - k * nx;
            x_update(e, k, i, om[L.rs + e], sv[L.oe + e]);
        }
        for (int cu = tid; cu < L.n_u; cu += NT) {
            const int k = divu<NUT>(L, cu), jj = cu - k * nu;
            u_update(cu, k, jj);
        }
    }
    __syncthreads();
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    auto row_update = [&](int r, double w, double &zv, double &yv) {
        double zt, lo, hi;
        if (r < L.rs) {                                   // dynamics
            const int k = divx<NXT>(L, r), i = r - k * nx;
            zt = -Tc[k * NB + i];
            if (k > 0) {
                const double *xp = Tc + (k - 1) * NB;
                const double *up = Tc + min(k - 1, L.Nc - 1) * NB + nx;
#pragma unroll
                for (int j = 0; j < (NXT ? NXT : 1); ++j) if (NXT) zt += Ad[i * nx + j] * xp[j];
                if (!NXT) for (int j = 0; j < nx; ++j) zt += Ad[i * nx + j] * xp[j];
#pragma unroll
                for (int j = 0; j < (NUT ? NUT : 1); ++j) if (NUT) zt += Bd[i * nu + j] * up[j];
                if (!NUT) for (int j = 0; j < nu; ++j) zt += Bd[i * nu + j] * up[j];
            }
            lo = hi = (r < nx) ? -x0s[r] : 0.0;
        } else if (r < L.ri) {                            // soft state box
            const int e = r - L.rs, k = divx<NXT>(L, e), i = e - k * nx;
            zt = Tc[k * NB + i] + W[r];
            lo = hot[L.oxmin + i]; hi = hot[L.oxmax + i];
        } else if (r < L.rdu) {                           // input box
            const int cu = r - L.ri, k = divu<NUT>(L, cu), jj = cu - k * nu;
            zt = Tc[k * NB + nx + jj];
            lo = hot[L.oumin + jj]; hi = hot[L.oumax + jj];
        } else {                                          // Delta-u rows
            const int rr = r - L.rdu, kk = divu<NUT>(L, rr), jj = rr - kk * nu;
            lo = hot[L.oDumin + jj]; hi = hot[L.oDumax + jj];
            if (rr < nu) { zt = Tc[nx + rr]; lo += um1s[jj]; hi += um1s[jj]; }
            else {
                const int cu = rr - nu, k = kk - 1;       // cu = k*nu + jj
                zt = -Tc[k * NB + nx + jj];
                if (cu + 1 < L.n_u) zt += (jj + 1 < nu) ? Tc[k * NB + nx + jj + 1] : Tc[(k + 1) * NB + nx];
            }
        }
        lo = lo < -QP_INFTY ? -QP_INFTY : lo;
        hi = hi > QP_INFTY ? QP_INFTY : hi;
        const double zr = alpha * zt + (1.0 - alpha) * zv;
        const double zn = fmin(fmax(zr + cc * yv / w, lo), hi);
        const double dy = (w / cc) * (zr - zn);
        yv += dy; zv = zn;
        if (keep_delta) dyg[r] = dy;
    };
    if (REGV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tid + NT * j;
            if (r < L.m) { double zv = Z[r], yv = Y[r]; row_update(r, h.om_r[j], zv, yv); Z[r] = zv; Y[r] = yv; W[r] = h.om_r[j] * zv - cc * yv; }
        }
    } else {
        for (int r = tid; r < L.m; r += NT) { double zv = Z[r], yv = Y[r]; const double w = om[r]; row_update(r, w, zv, yv); Z[r] = zv; Y[r] = yv; W[r] = w * zv - cc * yv; }
    }
    __syncthreads();
}

// `iters` ADMM iterations of this workgroup's instance.  Expects the hot model prefix and the step data in LDS
// (load_common) and, with LDSSTATE, X/Z/Y carved behind the common area.
template <int NB, bool LDSSTATE, int NXT, int NUT, bool BORDER>
__device__ __forceinline__ void admm_body(const Lay &L, const HotPtrs &P, Smem &S, double *X, double *Z, double *Y, double alpha, int iters) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    double *W = S.T, *Tc = S.T + L.m;
    if (LDSSTATE) {          // small-problem mode: the iterate x, z, y lives in LDS for the whole round
        for (int j = tid; j < L.n; j += NT) X[j] = gx[j];
        for (int r = tid; r < L.m; r += NT) { Z[r] = gz[r]; Y[r] = gy[r]; }
    } else { X = gx; Z = gz; Y = gy; }
    __syncthreads();
    cgdouble *gom = (cgdouble *)(P.omega + (size_t)b * L.m), *gsv = (cgdouble *)(P.s + (size_t)b * L.n), *gqv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double *F = P.F + (size_t)b * P.fsz;
    const double cc = P.c[b];
    HotRegs hr;
    if (LDSSTATE) load_hot_regs<NB, NXT, NUT>(L, gom, gsv, gqv, hr);
#ifndef MPCQP_ABL_NOPAR
    hot_rows_w<LDSSTATE>(L, gom, hr, cc, Z, Y, W);
#endif
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;         // the increments feed the infeasibility certificates of the check
        TICK_START
#ifndef MPCQP_ABL_NOPAR
        hot_rhs<NB, NXT, NUT, LDSSTATE>(L, S.hot, gom, gsv, gqv, hr, cc, X, Z, Y, W, Tc);
#endif
        TICK(0)
        BorderPtrs bp; bp.red = S.red;
        if (BORDER) {
            const size_t npb = (size_t)L.nu * L.N * L.NB;
            bp.Bb = (double *)P.Bb + blockIdx.x * npb; bp.Zb = (double *)P.Zb + blockIdx.x * npb; bp.Sig = (double *)P.Sig + (size_t)blockIdx.x * L.nu * L.nu;
        }
        if (BORDER) border_pre<NB>(L, bp.Bb, bp.Zb, bp.Sig, Tc, S.tv, S.red);
        kkt_core<NB>(core_args(L, opaque_ptr(F)), Tc);
        if (BORDER) border_post(L, NB, Tc, S.tv);
#ifndef MPCQP_ABL_NOPAR
        hot_update<NB, NXT, NUT, LDSSTATE>(L, S.hot, S.x0s, S.um1s, gom, gsv, hr, cc, alpha, X, Z, Y, W, Tc, keep_delta, dxg, dyg);
#endif
        TICK(4)
    }
    if (LDSSTATE) {
        for (int j = tid; j < L.n; j += NT) gx[j] = X[j];
        for (int r = tid; r < L.m; r += NT) { gz[r] = Z[r]; gy[r] = Y[r]; }
    }
}

// ------------------------------------------------------------------------------------------------
// Device-side receding-horizon loop (the caller pattern of examples/example_point_mass.py:88-101 and
// pyMPC/mpc.py:688-692):   for k in range(K):  u = K.output();  x = Ap x + Bp u + w_k;  K.update(x)
// One workgroup walks its own instance through all K steps -- output (mpc.py:271-336, u_failure = uref unless
// 'solved'), plant, QP refresh (mpc.py:386-454), warm-started solve -- with no host round trip and, unlike the
// per-step API, no batch-wide barrier per round: an instance that needs 50 iterations does not hold up one that
// needs 25.  The same kernel with nsteps = 0 is
// mpcqp_solve: begin, rounds of { admm, check } until this instance terminates -- no host loop, no batch barrier.
// ------------------------------------------------------------------------------------------------
struct RunArgs {
    int nsteps;                   // closed-loop steps (LOOP kernels); 0 = one solve of the current data (mpcqp_solve)
    int plain;                    // run exactly max_iter iterations, no termination test / rho adaptation (mpcqp_iterate)
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

struct RunSmem { Smem S; double *X, *Z, *Y; };
template <bool LDSSTATE>
__device__ __forceinline__ RunSmem run_smem(const Lay &L, const Ptrs &P) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    RunSmem r; double *p = sh; smem_common(L, P, p, r.S);
    r.X = r.Z = r.Y = nullptr;
    if (LDSSTATE) { r.X = carve(p, L.n); r.Z = carve(p, L.m); r.Y = carve(p, L.m); }
    return r;
}

template <int NB>
__device__ __noinline__ void run_factor_phase() {
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P;
    RunSmem r = run_smem<false>(L, P);                       // (the common LDS area comes first in both layouts)
    const int b = blockIdx.x;
    Ctx c{L, r.S.hot, P.model + (size_t)b * L.model_sz};
    factor_all<NB>(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, r.S.T, r.S.iflag,
                   border_ptrs(L, P, r.S.red));
}

template <int NB, bool LDSSTATE, int NXT, int NUT, bool BORDER>
__device__ __noinline__ void run_admm_phase(int iters) {
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P;
    RunSmem r = run_smem<LDSSTATE>(L, P);
    HotPtrs hp; hp.model = P.model; hp.step = P.step; hp.omega = P.omega; hp.s = P.s; hp.qv = P.qv; hp.F = P.F; hp.c = P.c;
    hp.Bb = P.Bb; hp.Zb = P.Zb; hp.Sig = P.Sig; hp.x = P.x; hp.z = P.z; hp.y = P.y; hp.dx = P.dx; hp.dy = P.dy; hp.fsz = P.fsz;
    admm_body<NB, LDSSTATE, NXT, NUT, BORDER>(L, hp, r.S, r.X, r.Z, r.Y, A.S.alpha, __builtin_amdgcn_readfirstlane(iters));
}

template <int NB, bool LDSSTATE>
__device__ __noinline__ void run_begin_phase(int plain) {
    const RunKArgs &A = run_kargs();
    RunSmem r = run_smem<LDSSTATE>(A.L, A.P);
    begin_body<NB>(A.L, A.P, A.S, r.S, __builtin_amdgcn_readfirstlane(plain));
}

template <int NB, bool LDSSTATE>
__device__ __noinline__ int run_check_phase(int iter, int mode) {
    const RunKArgs &A = run_kargs();
    RunSmem r = run_smem<LDSSTATE>(A.L, A.P);
    return check_body<NB>(A.L, A.P, A.S, r.S, __builtin_amdgcn_readfirstlane(iter), __builtin_amdgcn_readfirstlane(mode), r.X, r.Z, r.Y);
}

template <int NB, bool LDSSTATE, int NXT, int NUT, bool BORDER, bool LOOP>
__global__ __launch_bounds__(NT, (NB <= 16 ? 4 : 2)) void k_mpc_run(RunKArgs A_) {
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P; const RunArgs &R = A.R;
    RunSmem rs = run_smem<LDSSTATE>(L, P);
    Smem &S = rs.S;
    const int b = blockIdx.x, tid = threadIdx.x;
    double *step = P.step + (size_t)b * L.step_sz;
    load_common(L, P.model + (size_t)b * L.model_sz, step, S);
    const int nx = L.nx, nu = L.nu;
    const int nrun = LOOP ? R.nsteps : 1;        // LOOP = false: one solve of the current data (mpcqp_solve)
    for (int k = 0; k < nrun; ++k) {
        if (LOOP) {
            // scratch in the (idle) work area: un | xn | xt | ym | inn | xu, 32 doubles each (nx + nu <= 32)
            double *un = S.T, *xn = S.T + 32, *xt = S.T + 64, *ym = S.T + 96, *inn = S.T + 128, *xu = S.T + 160;
            const size_t kb = (size_t)k * R.batch + b;
            const int ny = R.ny;
            // ---- output(): first input of the current solution, or u_failure
            const int status = P.info[b].status;
            if (tid < nu) un[tid] = status == MPCQP_SOLVED ? P.xo[(size_t)b * L.n + L.ou + tid] : S.hot[L.ouref + tid];
            if (tid < nx) xt[tid] = ny ? R.x_true[(size_t)b * nx + tid] : S.x0s[tid];      // the plant state
            __syncthreads();
            if (ny && tid < ny) {                            // measurement y = C x + v and innovation y - C xhat
                const double *C = R.C + (size_t)b * ny * nx + (size_t)tid * nx;
                double y = R.v ? R.v[kb * ny + tid] : 0.0, yh = 0.0;
                for (int j = 0; j < nx; ++j) { y += C[j] * xt[j]; yh += C[j] * S.x0s[j]; }
                ym[tid] = y; inn[tid] = y - yh;
                if (R.y_traj) R.y_traj[kb * ny + tid] = y;
            }
            // ---- plant step
            if (tid < nx) {
                const double *Ap = R.Ap ? R.Ap + (size_t)b * nx * nx : S.hot + L.oAd;
                const double *Bp = R.Bp ? R.Bp + (size_t)b * nx * nu : S.hot + L.oBd;
                double v = R.w ? R.w[kb * nx + tid] : 0.0;
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) acc += Ap[tid * nx + j] * xt[j];
                for (int j = 0; j < nu; ++j) acc += Bp[tid * nu + j] * un[j];
                xn[tid] = acc + v;
                R.x_traj[kb * nx + tid] = xt[tid];
                if (ny) { R.x_true[(size_t)b * nx + tid] = xn[tid]; if (R.xhat_traj) R.xhat_traj[kb * nx + tid] = S.x0s[tid]; }
            }
            if (tid < nu) R.u_traj[kb * nu + tid] = un[tid];
            __syncthreads();
            if (ny) {                                        // KF.update(y): xhat[k|k] = xhat[k|k-1] + L (y - yhat);  KF.predict(u)
                if (tid < nx) {
                    const double *Lg = R.Lg + (size_t)b * nx * ny + (size_t)tid * ny;
                    double acc = S.x0s[tid];
                    for (int j = 0; j < ny; ++j) acc += Lg[j] * inn[j];
                    xu[tid] = acc;
                }
                __syncthreads();
                if (tid < nx) {
                    const double *Ad = S.hot + L.oAd, *Bd = S.hot + L.oBd;
                    double acc = 0.0;
                    for (int j = 0; j < nx; ++j) acc += Ad[tid * nx + j] * xu[j];
                    for (int j = 0; j < nu; ++j) acc += Bd[tid * nu + j] * un[j];
                    xn[tid] = acc;                           // xhat[k+1|k]: what the controller is updated with
                }
                __syncthreads();
            }
            // ---- update(x): new initial state, previous input (mpc.py:338-364) and, if given, reference
            if (tid < nx) { S.x0s[tid] = xn[tid]; step[tid] = xn[tid]; }
            if (tid < nu) { S.um1s[tid] = un[tid]; step[nx + tid] = un[tid]; }
            if (R.xref_traj) for (int i = tid; i < R.xref_blk; i += NT) step[nx + nu + i] = R.xref_traj[kb * R.xref_blk + i];
            __syncthreads();
        }
#ifdef MPCQP_RUN_TIMING
#define PHASE_CLOCK(i) { unsigned long long t_ = wall_clock64(); if (tid == 0) atomicAdd(&P.stats[4 + (i)], t_ - tphase); tphase = t_; }
        unsigned long long tphase = wall_clock64();
#else
#define PHASE_CLOCK(i)
#endif
        run_begin_phase<NB, LDSSTATE>(R.plain);
        __syncthreads();
        PHASE_CLOCK(0)
        int iter = 0, term = 0;
        while (!term) {
            const int nxt = next_stop(iter, R.max_iter, R.chk, R.rho_every);
            run_admm_phase<NB, LDSSTATE, NXT, NUT, BORDER>(nxt - iter);
            iter = nxt;
            __syncthreads();
            PHASE_CLOCK(1)
            term = run_check_phase<NB, LDSSTATE>(iter, stop_mode(iter, R.max_iter, R.chk, R.rho_every, R.plain != 0));
            __syncthreads();
            PHASE_CLOCK(2)
        }
        if (LOOP && tid == 0) {
            R.status_traj[(size_t)k * R.batch + b] = P.info[b].status;
            R.iter_traj[(size_t)k * R.batch + b] = iter;
        }
        __syncthreads();
    }
    if (LOOP && tid < nx) {
        const size_t e = ((size_t)R.nsteps * R.batch + b) * nx + tid;
        R.x_traj[e] = R.ny ? R.x_true[(size_t)b * nx + tid] : S.x0s[tid];
        if (R.ny && R.xhat_traj) R.xhat_traj[e] = S.x0s[tid];
    }
}

// ------------------------------------------------------------------------------------------------
// verification kernels
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(NT) void k_export(Lay L, Ptrs P, double *Pd, double *Ad_, double *q, double *l, double *u) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model};
    build_q(c, step, S.Qv);
    __syncthreads();
    if (Pd) { double *o = Pd + (size_t)b * L.n * L.n; for (int j = tid; j < L.n; j += NT) P_row(c, j, [&](double co, int idx) { o[(size_t)j * L.n + idx] = co; }); }
    if (Ad_) { double *o = Ad_ + (size_t)b * L.m * L.n; for (int r = tid; r < L.m; r += NT) A_row(c, r, [&](double co, int idx) { o[(size_t)r * L.n + idx] = co; }); }
    if (q) for (int j = tid; j < L.n; j += NT) q[(size_t)b * L.n + j] = (j < L.oe) ? S.Qv[j] : 0.0;
    if (l && u) for (int r = tid; r < L.m; r += NT) { double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi); l[(size_t)b * L.m + r] = lo; u[(size_t)b * L.m + r] = hi; }
}

template <int NB>
__global__ __launch_bounds__(NT) void k_kkt_solve(Lay L, Ptrs P, const double *rhs, double *sol) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model};
    kkt_solve<NB>(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, rhs + (size_t)b * L.n, S.T + L.m, sol + (size_t)b * L.n, border_ptrs(L, P, S.red), S.tv);
    (void)tid;
}

__global__ void k_gather_u0(Lay L, const double *xo, double *u0, int batch) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * L.nu) { int b = i / L.nu, j = i - b * L.nu; u0[i] = xo[(size_t)b * L.n + L.ou + j]; }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
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
    void *run_buf; size_t run_bytes;     // staging of mpcqp_mpc_run (disturbances, plant, trajectories)
    bool profiling;
    hipEvent_t ev0[MAXEV], ev1[MAXEV];   // ring of event pairs around the solve-kernel launches
    long long ev_count;
    double run_ms;
    long long run_launches;
    bool have_events;
};

extern "C" void mpcqp_default_settings(mpcqp_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
    s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 4000; s->check_termination = 25; s->scaling = 10;
    s->adaptive_rho = 1; s->adaptive_rho_interval = 0; s->warm_start = 1;
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

static Lay make_layout(int nx, int nu, int Np, int Nc) {
    Lay L; memset(&L, 0, sizeof(L));
    L.nx = nx; L.nu = nu; L.Np = Np; L.Nc = Nc; L.N = Np + 1; L.nb = nx + nu;
    L.n_x = L.N * nx; L.n_u = Nc * nu;
    L.n = 2 * L.n_x + L.n_u; L.m = 2 * L.n_x + L.n_u + (Nc + 1) * nu;
    L.ou = L.n_x; L.oe = L.n_x + L.n_u;
    L.rs = L.n_x; L.ri = 2 * L.n_x; L.rdu = 2 * L.n_x + L.n_u;
    L.NB = L.nb <= 16 ? 16 : 32;
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
    L.step_sz = nx + nu + L.N * nx;
    L.xref_rows = 1;
    L.fstage = 2 * L.NB * L.NB;
    L.tsz = (L.m + L.N * L.NB) > 6 * L.NB * L.NB ? (L.m + L.N * L.NB) : 6 * L.NB * L.NB;   // [W (m) | Tc (N*NB)] or factor workspace
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
    if (nx + nu > 32) return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_create: nx+nu > 32 is not implemented");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPCQP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MPCQP_ERR_ARG, "mpcqp_create: bad device index");
    HIPCHK(hipSetDevice(device));
    mpcqp_handle *h = new mpcqp_handle();
    h->device = device; h->batch = batch; h->is_setup = false; h->u0_dev = nullptr; h->run_buf = nullptr; h->run_bytes = 0;
    h->profiling = false; h->run_ms = 0.0; h->run_launches = 0; h->ev_count = 0; h->have_events = false;
    h->L = make_layout(nx, nu, Np, Nc);
    if (s) h->S = *s; else mpcqp_default_settings(&h->S);
    HIPCHK(hipStreamCreate(&h->stream));
    h->own_stream = true;
    for (int e = 0; e < MAXEV; ++e) { HIPCHK(hipEventCreate(&h->ev0[e])); HIPCHK(hipEventCreate(&h->ev1[e])); }
    h->have_events = true;
    const Lay &L = h->L;
    Ptrs &P = h->P; memset(&P, 0, sizeof(P));
    size_t B = (size_t)batch;
    P.fsz = (long long)L.N * L.fstage;
    int rc = 0;
    rc |= dalloc(h, &P.model, B * L.model_sz); rc |= dalloc(h, &P.step, B * L.step_sz);
    rc |= dalloc(h, &P.D, B * L.n); rc |= dalloc(h, &P.E, B * L.m); rc |= dalloc(h, &P.c, B);
    rc |= dalloc(h, &P.omega, B * L.m); rc |= dalloc(h, &P.s, B * L.n); rc |= dalloc(h, &P.rho, B);
    rc |= dalloc(h, &P.F, B * (size_t)P.fsz);
    rc |= dalloc(h, &P.x, B * L.n); rc |= dalloc(h, &P.z, B * L.m); rc |= dalloc(h, &P.y, B * L.m);
    rc |= dalloc(h, &P.xo, B * L.n); rc |= dalloc(h, &P.yo, B * L.m);
    rc |= dalloc(h, &P.dx, B * L.n); rc |= dalloc(h, &P.dy, B * L.m);
    rc |= dalloc(h, &P.qv, B * (size_t)(L.n_x + L.n_u));
    if (L.border) { rc |= dalloc(h, &P.Bb, B * (size_t)L.nu * L.N * L.NB); rc |= dalloc(h, &P.Zb, B * (size_t)L.nu * L.N * L.NB); rc |= dalloc(h, &P.Sig, B * (size_t)L.nu * L.nu); }
    rc |= dalloc(h, &P.Dt, B * L.n); rc |= dalloc(h, &P.Et, B * L.m);
    rc |= dalloc(h, &P.ctype, B * L.m); rc |= dalloc(h, &P.info, B); rc |= dalloc(h, &P.stats, 8);
        rc |= dalloc(h, &h->u0_dev, B * L.nu);
    if (rc) { mpcqp_destroy(h); return MPCQP_ERR_HIP; }
    h->smem_setup = sizeof(double) * (size_t)smem_common_doubles(L);
    size_t with_state = h->smem_setup + sizeof(double) * (size_t)(L.n + 2 * L.m);
    h->lds_state = with_state <= 40 * 1024 && L.m <= 4 * NT && L.N * L.NB <= 2 * NT;      // four workgroups per CU; larger problems keep the iterate in L2/HBM   // small problems: x in LDS, z/y in registers; else iterate in L2/HBM
    h->smem_solve = h->lds_state ? with_state : h->smem_setup;
    if (h->smem_solve > 160 * 1024) { mpcqp_destroy(h); return fail(MPCQP_ERR_UNSUPPORTED, "problem too large for one workgroup's LDS"); }
    *out = h;
    return MPCQP_OK;
}

extern "C" void mpcqp_destroy(mpcqp_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    if (h->run_buf) hipFree(h->run_buf);
    if (h->have_events) for (int e = 0; e < MAXEV; ++e) { hipEventDestroy(h->ev0[e]); hipEventDestroy(h->ev1[e]); }
    if (h->own_stream) hipStreamDestroy(h->stream);
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

// strided upload: src [batch][w] -> dst [batch][stride] at column offset off
static int put(mpcqp_handle *h, double *dst, int stride, int off, const double *src, int w) {
    HIPCHK(hipMemcpy2DAsync(dst + off, sizeof(double) * (size_t)stride, src, sizeof(double) * (size_t)w,
                            sizeof(double) * (size_t)w, (size_t)h->batch, hipMemcpyDefault, h->stream));
    return 0;
}

#define DISPATCH_NB(NBV, EXPR) switch (NBV) { \
    case 16: { constexpr int NB = 16; EXPR; } break; \
    default: { constexpr int NB = 32; EXPR; } break; }

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
    return 0;
}

extern "C" int mpcqp_setup(mpcqp_handle *h, const mpcqp_model *M, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h || !M || !x0 || !um1 || !xref) return fail(MPCQP_ERR_ARG, "mpcqp_setup: null argument");
    if (!M->Ad || !M->Bd || !M->Qx || !M->QxN || !M->Qu || !M->QDu || !M->xmin || !M->xmax || !M->umin || !M->umax ||
        !M->Dumin || !M->Dumax || !M->uref || !M->eps_feas) return fail(MPCQP_ERR_ARG, "mpcqp_setup: null model field");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; const int nx = L.nx, nu = L.nu, ms = L.model_sz;
    double *mb = h->P.model;
    int rc = 0;
    rc |= put(h, mb, ms, L.oAd, M->Ad, nx * nx); rc |= put(h, mb, ms, L.oBd, M->Bd, nx * nu);
    rc |= put(h, mb, ms, L.oxmin, M->xmin, nx); rc |= put(h, mb, ms, L.oxmax, M->xmax, nx);
    rc |= put(h, mb, ms, L.oumin, M->umin, nu); rc |= put(h, mb, ms, L.oumax, M->umax, nu);
    rc |= put(h, mb, ms, L.oDumin, M->Dumin, nu); rc |= put(h, mb, ms, L.oDumax, M->Dumax, nu);
    rc |= put(h, mb, ms, L.ouref, M->uref, nu); rc |= put(h, mb, ms, L.oeps, M->eps_feas, 1);
    rc |= put(h, mb, ms, L.oQx, M->Qx, nx * nx); rc |= put(h, mb, ms, L.oQxN, M->QxN, nx * nx);
    rc |= put(h, mb, ms, L.oQu, M->Qu, nu * nu); rc |= put(h, mb, ms, L.oQDu, M->QDu, nu * nu);
    if (rc) return MPCQP_ERR_HIP;
    if ((rc = step_upload(h, x0, um1, xref, xref_rows))) return rc;
    DISPATCH_NB(L.NB, {
        if (set_smem(k_setup<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_setup<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, h->S);
    });
    HIPCHK(hipGetLastError());
    h->is_setup = true;
    return MPCQP_OK;
}

extern "C" int mpcqp_update(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_update before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    return step_upload(h, x0, um1, xref, xref_rows);
}

extern "C" int mpcqp_update_settings(mpcqp_handle *h, const mpcqp_settings *s) {
    if (!h || !s) return fail(MPCQP_ERR_ARG, "null argument");
    double rho = h->S.rho, sigma = h->S.sigma; int scaling = h->S.scaling;
    h->S = *s;
    h->S.rho = rho; h->S.sigma = sigma; h->S.scaling = scaling;   // fixed at setup (they shape the factorization)
    return MPCQP_OK;
}

template <int NB, bool LDSS, int NXT, int NUT, bool BORDER>
static int launch_run_t(mpcqp_handle *h, const RunArgs &R) {
    RunKArgs A; A.L = h->L; A.P = h->P; A.S = h->S; A.R = R;
    if (R.nsteps > 0) {
        if (set_smem(k_mpc_run<NB, LDSS, NXT, NUT, BORDER, true>, h->smem_solve)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL((k_mpc_run<NB, LDSS, NXT, NUT, BORDER, true>), dim3(h->batch), dim3(NT), h->smem_solve, h->stream, A);
    } else {
        if (set_smem(k_mpc_run<NB, LDSS, NXT, NUT, BORDER, false>, h->smem_solve)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL((k_mpc_run<NB, LDSS, NXT, NUT, BORDER, false>), dim3(h->batch), dim3(NT), h->smem_solve, h->stream, A);
    }
    return 0;
}
template <int NB, bool LDSS>
static int launch_run_generic(mpcqp_handle *h, const RunArgs &R) {
    return h->L.border ? launch_run_t<NB, LDSS, 0, 0, true>(h, R) : launch_run_t<NB, LDSS, 0, 0, false>(h, R);
}

// One launch of the solve / closed-loop kernel on the handle's stream (asynchronous).  Specialisations with
// compile-time nx, nu for the BASELINE configurations; generic kernels otherwise.
static int launch_run(mpcqp_handle *h, RunArgs R, int plain_iters) {
    const Lay &L = h->L; const mpcqp_settings &S = h->S;
    R.plain = plain_iters > 0;
    R.max_iter = R.plain ? plain_iters : S.max_iter;
    R.chk = R.plain ? 0 : S.check_termination;
    R.rho_every = (!R.plain && S.adaptive_rho) ? (S.adaptive_rho_interval ? S.adaptive_rho_interval : (R.chk ? 4 * R.chk : 100)) : 0;
    R.batch = h->batch;
    const int e = h->ev_count % MAXEV;
    if (h->profiling) {
        if (h->ev_count >= MAXEV) {                 // ring full: bank the oldest pair first
            float ms = 0.f; HIPCHK(hipEventSynchronize(h->ev1[e])); HIPCHK(hipEventElapsedTime(&ms, h->ev0[e], h->ev1[e]));
            h->run_ms += ms; h->run_launches += 1;
        }
        HIPCHK(hipEventRecord(h->ev0[e], h->stream));
    }
    int rc;
    if (L.NB == 16 && h->lds_state && L.nx == 12 && L.nu == 4 && !L.border) rc = launch_run_t<16, true, 12, 4, false>(h, R);
    else if (L.NB == 32 && !h->lds_state && L.nx == 20 && L.nu == 8 && !L.border) rc = launch_run_t<32, false, 20, 8, false>(h, R);
    else if (L.NB == 16) rc = h->lds_state ? launch_run_generic<16, true>(h, R) : launch_run_generic<16, false>(h, R);
    else rc = h->lds_state ? launch_run_generic<32, true>(h, R) : launch_run_generic<32, false>(h, R);
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

static int launch_solve(mpcqp_handle *h, int plain_iters) {
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "solve before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    RunArgs R; memset(&R, 0, sizeof(R));
    return launch_run(h, R, plain_iters);
}

extern "C" int mpcqp_solve(mpcqp_handle *h) { if (!h) return fail(MPCQP_ERR_ARG, "null handle"); return launch_solve(h, 0); }
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
    if (ny < 0 || ny > 32) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: ny must be in 0..32");
    if (ny && (!io->C || !io->Lgain || !io->x_true)) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: output feedback needs C, Lgain and x_true");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L;
    const size_t B = (size_t)h->batch, K = (size_t)nsteps, nx = L.nx, nu = L.nu;
    const size_t xblk = (size_t)L.xref_rows * nx;
    // one staging block on the device; inputs are copied in, outputs copied out (host or device pointers alike)
    struct Part { const void *src; void *dst; size_t bytes; size_t off; };
    Part parts[16]; int np = 0; size_t total = 0;
    auto add = [&](const void *src, void *dst, size_t bytes) { parts[np] = Part{src, dst, bytes, total}; total += (bytes + 15) & ~size_t(15); return np++; };
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
    auto dev = [&](int i) -> void * { return parts[i].bytes ? base + parts[i].off : nullptr; };
    for (int i = 0; i < np; ++i)
        if (parts[i].src && parts[i].bytes) HIPCHK(hipMemcpyAsync(dev(i), parts[i].src, parts[i].bytes, hipMemcpyDefault, h->stream));
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
        if (parts[i].dst && parts[i].bytes && get(h, parts[i].dst, dev(i), parts[i].bytes)) return MPCQP_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_mpc_run(mpcqp_handle *h, int nsteps, const double *w, const double *Ap, const double *Bp,
                             double *x_traj, double *u_traj, int32_t *status_traj, int32_t *iter_traj) {
    mpcqp_loop io; memset(&io, 0, sizeof(io));
    io.w = w; io.Ap = Ap; io.Bp = Bp; io.x_traj = x_traj; io.u_traj = u_traj; io.status_traj = status_traj; io.iter_traj = iter_traj;
    return mpcqp_mpc_loop(h, nsteps, &io);
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
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

extern "C" int mpcqp_get_stats(mpcqp_handle *h, uint64_t *out4, int reset) {
    if (!h || !out4) return fail(MPCQP_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out4, h->P.stats, 4 * sizeof(uint64_t), hipMemcpyDefault, h->stream));
#ifdef MPCQP_RUN_TIMING
    { uint64_t t[4]; hipMemcpy(t, h->P.stats + 4, sizeof(t), hipMemcpyDeviceToHost); fprintf(stderr, "phase wall-clock ticks: begin %llu admm %llu check %llu\n", (unsigned long long)t[0], (unsigned long long)t[1], (unsigned long long)t[2]);
      unsigned long long g[16]; hipMemcpyFromSymbol(g, HIP_SYMBOL(g_ticks), sizeof(g)); unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), z, sizeof(z));
      fprintf(stderr, "iteration ticks: rhs %llu fwd %llu sinv %llu bwd %llu (kkt tail in update) update %llu\n", g[0], g[1], g[2], g[3], g[4]); }
#endif
    if (reset) HIPCHK(hipMemsetAsync(h->P.stats, 0, 8 * sizeof(uint64_t), h->stream));
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
        *nnzL = full + sub + border + 2 * (int64_t)L.n_x;
    }
    return MPCQP_OK;
}

extern "C" int mpcqp_warm_start(mpcqp_handle *h, const double *x, const double *y) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "warm_start before setup");
    HIPCHK(hipSetDevice(h->device));
    size_t B = (size_t)h->batch;
    if (x) { HIPCHK(hipMemcpyAsync(h->P.x, x, B * h->L.n * sizeof(double), hipMemcpyDefault, h->stream)); }
    if (y) { HIPCHK(hipMemcpyAsync(h->P.y, y, B * h->L.m * sizeof(double), hipMemcpyDefault, h->stream)); }
    return MPCQP_OK;
}

extern "C" int mpcqp_export_qp(mpcqp_handle *h, double *Pm, double *Am, double *q, double *l, double *u) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "export before setup");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L; size_t B = (size_t)h->batch;
    double *dP = nullptr, *dA = nullptr, *dq = nullptr, *dl = nullptr, *du = nullptr;
    if (Pm) { HIPCHK(hipMalloc((void **)&dP, B * L.n * L.n * sizeof(double))); HIPCHK(hipMemsetAsync(dP, 0, B * L.n * L.n * sizeof(double), h->stream)); }
    if (Am) { HIPCHK(hipMalloc((void **)&dA, B * L.m * L.n * sizeof(double))); HIPCHK(hipMemsetAsync(dA, 0, B * L.m * L.n * sizeof(double), h->stream)); }
    if (q) HIPCHK(hipMalloc((void **)&dq, B * L.n * sizeof(double)));
    if (l && u) { HIPCHK(hipMalloc((void **)&dl, B * L.m * sizeof(double))); HIPCHK(hipMalloc((void **)&du, B * L.m * sizeof(double))); }
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
    hipFree(dP); hipFree(dA); hipFree(dq); hipFree(dl); hipFree(du);
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
    HIPCHK(hipMalloc((void **)&dr, bytes)); HIPCHK(hipMalloc((void **)&ds, bytes));
    HIPCHK(hipMemcpyAsync(dr, rhs, bytes, hipMemcpyDefault, h->stream));
    DISPATCH_NB(L.NB, {
        if (set_smem(k_kkt_solve<NB>, h->smem_setup)) return MPCQP_ERR_HIP;
        hipLaunchKernelGGL(k_kkt_solve<NB>, dim3(h->batch), dim3(NT), h->smem_setup, h->stream, h->L, h->P, dr, ds);
    });
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sol, ds, bytes, hipMemcpyDefault, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(dr); hipFree(ds);
    return MPCQP_OK;
}

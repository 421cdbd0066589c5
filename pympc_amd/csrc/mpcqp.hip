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
//     and is factored by a block Cholesky whose factor streams from HBM/L2 every iteration;
//   - one 256-thread workgroup owns one instance for the whole solve: iterate in LDS, wave 0 runs the
//     sequential block forward/backward substitution, all waves run the stage-parallel parts.
//
// FP64 throughout.  No CPU fallback exists in this library.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"

#define NT 256
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
struct Lay {
    int nx, nu, Np, Nc, N, nb, n, m, n_x, n_u, ou, oe, rs, ri, rdu;
    int NB;                       // padded stage block size (4, 8, 16, 32)
    float rnx, rnu;               // reciprocals for cheap index division
    // model blob: hot prefix [Ad|Bd|xmin|xmax|umin|umax|Dumin|Dumax|uref|eps_feas] then [Qx|QxN|Qu|QDu]
    int oAd, oBd, oxmin, oxmax, oumin, oumax, oDumin, oDumax, ouref, oeps, hot_sz;
    int oQx, oQxN, oQu, oQDu, model_sz;
    int step_sz;                  // [x0 | um1 | xref(N*nx)]
    int xref_rows;                // 1 or N
    int fstage;                   // doubles per factor stage: Lsub (NB*NB) + Linv (NB*NB)
    int tsz;                      // LDS work vector length: max(m, 4*NB*NB)
};

struct Ptrs {
    double *model, *step;
    double *D, *E, *c, *omega, *s, *rho;
    double *F;
    double *x, *z, *y;            // iterate (unscaled units)
    double *xo, *yo;              // reported solution
    double *dx, *dy, *rg;         // scratch: last increments, rhs
    double *Dt, *Et;              // Ruiz temporaries
    int *ctype;
    unsigned long long *stats;    // [0] ADMM iterations, [1] residual evaluations, [2] refactorizations, [3] instance-solves
    mpcqp_info *info;
    long long fsz;                // factor doubles per instance
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
    for (int i = 0; i < KMAX; ++i) vmax[i] = fmax(fmax(red[i], red[K + i]), fmax(red[2 * K + i], red[3 * K + i]));
#pragma unroll
    for (int i = 0; i < KSUM; ++i) vsum[i] = (red[KMAX + i] + red[K + KMAX + i]) + (red[2 * K + KMAX + i] + red[3 * K + KMAX + i]);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Reduced KKT matrix  K = c P + diag(s) + A' diag(omega) A  with eps eliminated: stage blocks.
// Stage k holds v_k = (x_k, u_k) (u absent in the last stage); blocks are NB x NB, identity padded.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double kkt_diag_entry(const Ctx &c, const double *om, const double *sv, double cc, int k, int a, int b) {
    const Lay &L = c.L;
    const int nbk = (k < L.Nc) ? L.nb : L.nx;
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
    const int nbk = (k < L.Nc) ? L.nb : L.nx;
    const int nbn = (k + 1 < L.Nc) ? L.nb : L.nx;
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

// Block Cholesky of the block-tridiagonal K.  Per stage k the factor stores
//   Lsub_k = L_{k,k-1}  (NB x NB, zero for k = 0)   and   Linv_k = L_kk^{-1} (lower triangular).
// W: LDS workspace of 4*NB*NB doubles.  Returns (uniformly) 0, or 1 if a pivot was not positive.
template <int NB>
__device__ int factor_all(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag) {
    const Lay &L = c.L;
    double *S = W, *Cm = W + NB * NB, *Ls = W + 2 * NB * NB, *Li = W + 3 * NB * NB;
    const int tid = threadIdx.x;
    if (tid == 0) *iflag = 0;
    for (int k = 0; k < L.N; ++k) {
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {
            int a = e / NB, b = e % NB;
            double v = kkt_diag_entry(c, om, sv, cc, k, a, b);
            if (k > 0) { double acc = 0.0; for (int l = 0; l < NB; ++l) acc += Ls[a * NB + l] * Ls[b * NB + l]; v -= acc; }
            S[e] = v;
            Cm[e] = (k < L.N - 1) ? kkt_sub_entry(c, om, cc, k, a, b) : 0.0;
            F[(size_t)k * L.fstage + e] = (k > 0) ? Ls[e] : 0.0;
            Li[e] = 0.0;
        }
        __syncthreads();
        // right-looking Cholesky of S (lower triangle)
        for (int j = 0; j < NB; ++j) {
            double d = S[j * NB + j];
            if (d <= 0.0) { if (tid == 0) *iflag = 1; d = 1e-300; }
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
        // Linv = S^{-1}: one thread per column, forward substitution on the identity
        if (tid < NB) {
            const int col = tid;
            Li[col * NB + col] = 1.0 / S[col * NB + col];
            for (int i = col + 1; i < NB; ++i) {
                double acc = 0.0;
                for (int l = col; l < i; ++l) acc += S[i * NB + l] * Li[l * NB + col];
                Li[i * NB + col] = -acc / S[i * NB + i];
            }
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += NT) {
            int a = e / NB, b = e % NB;
            F[(size_t)k * L.fstage + NB * NB + e] = Li[e];
            double acc = 0.0;                         // Lsub_{k+1} = K_{k+1,k} Linv_k'
            for (int l = 0; l <= b; ++l) acc += Cm[a * NB + l] * Li[b * NB + l];
            Ls[e] = acc;
        }
    }
    __syncthreads();
    return *iflag;
}

// ------------------------------------------------------------------------------------------------
// Block forward/backward substitution, executed by ONE wave (no workgroup barriers inside).
// T holds the right-hand side / solution of the x and u variables in the reference's flat layout.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NB>
__device__ void kkt_chain_solve(const Ctx &c, const double *F, double *T, double *tv) {
    constexpr int CPL = (NB * NB / 64) > 0 ? (NB * NB / 64) : 1;   // matrix entries per lane per row
    constexpr int LPR = NB / CPL;                                  // lanes per row
    const Lay &L = c.L;
    const int lane = threadIdx.x & 63;
    const bool act = lane < NB * LPR;
    auto vaddr = [&](int k, int a) { return a < L.nx ? k * L.nx + a : L.ou + k * L.nu + (a - L.nx); };

    // ---- forward: y_k = Linv_k (b_k - Lsub_k y_{k-1}) ; lane = (row r, column group q)
    {
        const int r = lane / LPR, q = lane % LPR;
        double m1[CPL], m2[CPL], n1[CPL], n2[CPL];
        const double *Fk = F;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { m1[j] = act ? Fk[r * NB + q * CPL + j] : 0.0; m2[j] = act ? Fk[NB * NB + r * NB + q * CPL + j] : 0.0; }
        for (int k = 0; k < L.N; ++k) {
            const int nbk = (k < L.Nc) ? L.nb : L.nx;
            if (k + 1 < L.N) {                        // prefetch the next stage's blocks
                const double *Fn = F + (size_t)(k + 1) * L.fstage;
#pragma unroll
                for (int j = 0; j < CPL; ++j) { n1[j] = act ? Fn[r * NB + q * CPL + j] : 0.0; n2[j] = act ? Fn[NB * NB + r * NB + q * CPL + j] : 0.0; }
            }
            double s1 = 0.0;
            if (k > 0) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) { int a = q * CPL + j; double yv = (a < L.nb) ? T[vaddr(k - 1, a)] : 0.0; s1 += m1[j] * yv; }
                for (int o = 1; o < LPR; o <<= 1) s1 += __shfl_xor(s1, o);
            }
            double t = ((act && r < nbk) ? T[vaddr(k, r)] : 0.0) - s1;
            if (act && q == 0) tv[r] = t;
            wave_lds_sync();
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < CPL; ++j) s2 += m2[j] * tv[(q * CPL + j) % NB];
            for (int o = 1; o < LPR; o <<= 1) s2 += __shfl_xor(s2, o);
            if (act && q == 0 && r < nbk) T[vaddr(k, r)] = s2;
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < CPL; ++j) { m1[j] = n1[j]; m2[j] = n2[j]; }
        }
    }
    // ---- backward: x_k = Linv_k' (y_k - Lsub_{k+1}' x_{k+1}) ; lane = (row group q, column cidx)
    {
        const int q = lane / NB, cidx = lane % NB;
        double m1[CPL], m2[CPL], n1[CPL], n2[CPL];
        {
            const double *Fk = F + (size_t)(L.N - 1) * L.fstage;
#pragma unroll
            for (int j = 0; j < CPL; ++j) { m1[j] = 0.0; m2[j] = act ? Fk[NB * NB + (q * CPL + j) * NB + cidx] : 0.0; }
        }
        for (int k = L.N - 1; k >= 0; --k) {
            const int nbk = (k < L.Nc) ? L.nb : L.nx;
            if (k > 0) {                               // prefetch: Lsub_k (used at stage k-1) and Linv_{k-1}
                const double *Fk = F + (size_t)k * L.fstage, *Fp = F + (size_t)(k - 1) * L.fstage;
#pragma unroll
                for (int j = 0; j < CPL; ++j) { n1[j] = act ? Fk[(q * CPL + j) * NB + cidx] : 0.0; n2[j] = act ? Fp[NB * NB + (q * CPL + j) * NB + cidx] : 0.0; }
            }
            double s1 = 0.0;
            if (k < L.N - 1) {
                const int nbn = (k + 1 < L.Nc) ? L.nb : L.nx;
#pragma unroll
                for (int j = 0; j < CPL; ++j) { int a = q * CPL + j; double xv = (a < nbn) ? T[vaddr(k + 1, a)] : 0.0; s1 += m1[j] * xv; }
                for (int o = NB; o < NB * LPR; o <<= 1) s1 += __shfl_xor(s1, o);
            }
            double t = ((act && cidx < nbk) ? T[vaddr(k, cidx)] : 0.0) - s1;
            if (act && q == 0) tv[cidx] = t;
            wave_lds_sync();
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < CPL; ++j) s2 += m2[j] * tv[(q * CPL + j) % NB];
            for (int o = NB; o < NB * LPR; o <<= 1) s2 += __shfl_xor(s2, o);
            if (act && q == 0 && cidx < nbk) T[vaddr(k, cidx)] = s2;
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < CPL; ++j) { m1[j] = n1[j]; m2[j] = n2[j]; }
        }
    }
}

// Solve K xt = rhs for all n variables.  On entry rg (global scratch) holds rhs; on exit T holds xt.
template <int NB>
__device__ void kkt_solve(const Ctx &c, const double *om, const double *sv, double cc, const double *F,
                          const double *rg, double *T, double *tv) {
    const Lay &L = c.L;
    const double cef = cc * c.eps_feas();
    for (int e = threadIdx.x; e < L.n_x; e += NT) {      // eliminate eps_e against x_e
        double ws = om[L.rs + e];
        double kap = cef + sv[L.oe + e] + ws;
        double te = rg[L.oe + e] / kap;
        T[L.oe + e] = te;
        T[e] = rg[e] - ws * te;
    }
    for (int j = threadIdx.x; j < L.n_u; j += NT) T[L.ou + j] = rg[L.ou + j];
    __syncthreads();
    if (threadIdx.x < 64) kkt_chain_solve<NB>(c, F, T, tv);
    __syncthreads();
    for (int e = threadIdx.x; e < L.n_x; e += NT) {
        double ws = om[L.rs + e];
        double kap = cef + sv[L.oe + e] + ws;
        T[L.oe + e] -= (ws / kap) * T[e];
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

// Shared prologue: stage the hot model prefix and the step data in LDS.
struct Smem {
    double *T, *Qv, *hot, *x0s, *um1s, *red, *tv;
    int *iflag;
};
__device__ __forceinline__ double *carve(double *&p, int n) { double *r = p; p += n; return r; }
__device__ void smem_common(const Lay &L, double *&p, Smem &S) {
    S.T = carve(p, L.tsz);
    S.Qv = carve(p, L.n_x + L.n_u);
    S.hot = carve(p, L.hot_sz);
    S.x0s = carve(p, L.nx);
    S.um1s = carve(p, L.nu);
    S.red = carve(p, 64);
    S.tv = carve(p, 32);
    S.iflag = (int *)carve(p, 2);
}
__host__ __device__ inline int smem_common_doubles(const Lay &L) { return L.tsz + L.n_x + L.n_u + L.hot_sz + L.nx + L.nu + 64 + 32 + 2; }

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
    double *p = sh; Smem S; smem_common(L, p, S);
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
    int bad = factor_all<NB>(c, om, sv, cc, P.F + (size_t)b * P.fsz, S.T, S.iflag);
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
// solve kernel: the whole warm-started ADMM solve of one instance in one persistent workgroup.
// ------------------------------------------------------------------------------------------------
template <int NB, bool LDSSTATE>
__global__ __launch_bounds__(NT, (NB <= 16 ? 4 : 2)) void k_solve(Lay L, Ptrs P, mpcqp_settings S_, int plain_iters) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, p, S);
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    double *X, *Z, *Y;
    if (LDSSTATE) { X = carve(p, L.n); Z = carve(p, L.m); Y = carve(p, L.m); }
    else { X = gx; Z = gz; Y = gy; }
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model};
    build_q(c, step, S.Qv);

    double *om = P.omega + (size_t)b * L.m, *sv = P.s + (size_t)b * L.n;
    const double *D = P.D + (size_t)b * L.n, *E = P.E + (size_t)b * L.m;
    double *F = P.F + (size_t)b * P.fsz;
    double *rg = P.rg + (size_t)b * L.n, *dxg = P.dx + (size_t)b * L.n, *dyg = P.dy + (size_t)b * L.m;
    int *ctp = P.ctype + (size_t)b * L.m;
    const double cc = P.c[b];
    double rho = P.rho[b];
    const double alpha = S_.alpha, sigma = S_.sigma;
    (void)sigma;
    const bool plain = plain_iters > 0;
    const int max_iter = plain ? plain_iters : S_.max_iter;
    const int chk_every = plain ? 0 : S_.check_termination;
    int rho_every = 0;
    if (!plain && S_.adaptive_rho) rho_every = S_.adaptive_rho_interval ? S_.adaptive_rho_interval : (S_.check_termination ? 4 * S_.check_termination : 100);

    // ---- prologue: iterate, constraint types (bounds may have changed since the last factorization)
    if (LDSSTATE) {
        const bool ws = S_.warm_start || plain;
        for (int j = tid; j < L.n; j += NT) X[j] = ws ? gx[j] : 0.0;
        for (int r = tid; r < L.m; r += NT) { Z[r] = ws ? gz[r] : 0.0; Y[r] = ws ? gy[r] : 0.0; }
    } else if (!(S_.warm_start || plain)) {
        for (int j = tid; j < L.n; j += NT) X[j] = 0.0;
        for (int r = tid; r < L.m; r += NT) { Z[r] = 0.0; Y[r] = 0.0; }
    }
    {
        int changed = 0;
        for (int r = tid; r < L.m; r += NT) {
            double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
            int t = row_type(E[r], lo, hi);
            if (t != ctp[r]) { changed = 1; ctp[r] = t; om[r] = row_rho(t, rho) * E[r] * E[r]; }
        }
        changed = __syncthreads_or(changed);
        if (changed) factor_all<NB>(c, om, sv, cc, F, S.T, S.iflag);
    }
    __syncthreads();

    int status = MPCQP_UNSOLVED, iter = 0, rho_updates = 0, n_info = 0;
    double obj_val = 0.0, pri_res = 0.0, dua_res = 0.0;
    bool have_info = false;

    // residuals / objective / scaled norms for the rho estimate: OSQP's update_info + compute_rho_estimate
    double nrm[12];
    auto update_info = [&]() {
        // vmax: 0 pri, 1 |Ax|, 2 |z|, 3 dua, 4 |Px|, 5 |A'y|, 6 |q|; scaled: 7 pri, 8 max(|EAx|,|Ez|), 9 dua, 10 max(|cD(..)|)
        double vmax[11], vsum[1] = {0.0};
#pragma unroll
        for (int i = 0; i < 11; ++i) vmax[i] = 0.0;
        for (int r = tid; r < L.m; r += NT) {
            double ax = 0.0;
            A_row(c, r, [&](double co, int idx) { ax += co * X[idx]; });
            double z = Z[r], d = ax - z, e = E[r];
            vmax[0] = fmax(vmax[0], fabs(d)); vmax[1] = fmax(vmax[1], fabs(ax)); vmax[2] = fmax(vmax[2], fabs(z));
            vmax[7] = fmax(vmax[7], fabs(e * d)); vmax[8] = fmax(vmax[8], fmax(fabs(e * ax), fabs(e * z)));
        }
        for (int j = tid; j < L.n; j += NT) {
            double px = 0.0, aty = 0.0;
            P_row(c, j, [&](double co, int idx) { px += co * X[idx]; });
            AT_row(c, j, [&](double co, int row) { aty += co * Y[row]; });
            double qj = (j < L.oe) ? S.Qv[j] : 0.0, xj = X[j];
            double d = px + qj + aty, cd = cc * D[j];
            vmax[3] = fmax(vmax[3], fabs(d)); vmax[4] = fmax(vmax[4], fabs(px)); vmax[5] = fmax(vmax[5], fabs(aty)); vmax[6] = fmax(vmax[6], fabs(qj));
            vmax[9] = fmax(vmax[9], fabs(cd * d));
            vmax[10] = fmax(vmax[10], fmax(fabs(cd * qj), fmax(fabs(cd * aty), fabs(cd * px))));
            vsum[0] += xj * (0.5 * px + qj);
        }
        block_reduce<11, 1>(vmax, vsum, S.red);
#pragma unroll
        for (int i = 0; i < 11; ++i) nrm[i] = vmax[i];
        obj_val = vsum[0]; pri_res = vmax[0]; dua_res = vmax[3];
        have_info = true; ++n_info;
    };
    auto rho_estimate = [&]() {
        double pri = nrm[7] / (nrm[8] + 1e-10), dua = nrm[9] / (nrm[10] + 1e-10);
        double r = rho * sqrt(pri / (dua + 1e-10));
        return fmin(fmax(r, RHO_MIN), RHO_MAX);
    };
    // OSQP's infeasibility certificates (paper section 3.5) on the last increments, in unscaled terms.
    auto primal_infeasible = [&](double eps) -> bool {
        // v = c * delta_y (= E * scaled delta_y), projected on the polar of the recession cone of [l,u]
        double vmax[1] = {0.0}, vsum[1] = {0.0};
        for (int r = tid; r < L.m; r += NT) {
            double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
            double e = E[r], v = cc * dyg[r];
            if (e * hi > QP_INFTY * MIN_SCALING) { if (e * lo < -QP_INFTY * MIN_SCALING) v = 0.0; else v = fmin(v, 0.0); }
            else if (e * lo < -QP_INFTY * MIN_SCALING) v = fmax(v, 0.0);
            S.T[r] = v;
            vmax[0] = fmax(vmax[0], fabs(v));
            vsum[0] += hi * fmax(v, 0.0) + lo * fmin(v, 0.0);
        }
        block_reduce<1, 1>(vmax, vsum, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vsum[0] < -eps * nd)) return false;
        double amax[1] = {0.0}, dummy[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double a = 0.0; AT_row(c, j, [&](double co, int row) { a += co * S.T[row]; });
            amax[0] = fmax(amax[0], fabs(a));
        }
        block_reduce<1, 1>(amax, dummy, S.red);
        return amax[0] < eps * nd;
    };
    auto dual_infeasible = [&](double eps) -> bool {
        double vmax[1] = {0.0}, vsum[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double d = dxg[j];
            vmax[0] = fmax(vmax[0], fabs(d));
            vsum[0] += ((j < L.oe) ? S.Qv[j] : 0.0) * d;
        }
        block_reduce<1, 1>(vmax, vsum, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vsum[0] < -eps * nd)) return false;
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

    bool done = false, can_check = false;
    for (iter = 1; iter <= max_iter; ++iter) {
        can_check = chk_every && (iter % chk_every == 0);
        const bool rho_now = rho_every && (iter % rho_every == 0);
        const bool keep_delta = can_check || iter == max_iter;
        // (1) w = omega z - c y
        for (int r = tid; r < L.m; r += NT) S.T[r] = om[r] * Z[r] - cc * Y[r];
        __syncthreads();
        // (2) rhs = s x - c q + A' w
        for (int j = tid; j < L.n; j += NT) {
            double acc = sv[j] * X[j] - ((j < L.oe) ? cc * S.Qv[j] : 0.0);
            AT_row(c, j, [&](double co, int row) { acc += co * S.T[row]; });
            rg[j] = acc;
        }
        __syncthreads();
        // (3) xt = K^-1 rhs
        kkt_solve<NB>(c, om, sv, cc, F, rg, S.T, S.tv);
        // (4) zt = A xt, relaxation, projection, dual update
        for (int r = tid; r < L.m; r += NT) {
            double zt = 0.0;
            A_row(c, r, [&](double co, int idx) { zt += co * S.T[idx]; });
            double zr = alpha * zt + (1.0 - alpha) * Z[r];
            double lo, hi; row_bounds(c, S.x0s, S.um1s, r, lo, hi);
            double w = om[r], yr = Y[r];
            double zn = fmin(fmax(zr + cc * yr / w, lo), hi);
            double dy = (w / cc) * (zr - zn);
            Y[r] = yr + dy; Z[r] = zn;
            if (keep_delta) dyg[r] = dy;
        }
        for (int j = tid; j < L.n; j += NT) {
            double xo = X[j];
            double xn = alpha * S.T[j] + (1.0 - alpha) * xo;
            X[j] = xn;
            if (keep_delta) dxg[j] = xn - xo;
        }
        __syncthreads();
        if (can_check) {
            update_info();
            if (check_termination(false)) { done = true; break; }
        }
        if (rho_now) {
            if (!can_check) update_info();
            double rn = rho_estimate();
            if (rn > rho * S_.adaptive_rho_tolerance || rn < rho / S_.adaptive_rho_tolerance) {
                rho = rn;
                for (int r = tid; r < L.m; r += NT) om[r] = row_rho(ctp[r], rho) * E[r] * E[r];
                __syncthreads();
                factor_all<NB>(c, om, sv, cc, F, S.T, S.iflag);
                ++rho_updates;
            }
        }
    }
    if (!done) {
        iter = max_iter;
        if (!can_check) update_info();
        if (plain) status = MPCQP_UNSOLVED;
        else if (!check_termination(true)) status = MPCQP_MAX_ITER_REACHED;
    }
    (void)have_info;

    // ---- epilogue: solution, iterate for the next warm start
    const bool has_sol = !(status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                           status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX);
    double *xo = P.xo + (size_t)b * L.n, *yo = P.yo + (size_t)b * L.m;
    for (int j = tid; j < L.n; j += NT) { double v = X[j]; xo[j] = has_sol ? v : NAN; gx[j] = has_sol ? v : 0.0; }
    for (int r = tid; r < L.m; r += NT) { double v = Y[r], zz = Z[r]; yo[r] = has_sol ? v : NAN; gy[r] = has_sol ? v : 0.0; gz[r] = has_sol ? zz : 0.0; }
    if (tid == 0) {
        mpcqp_info inf; inf.status = status; inf.iter = iter; inf.rho_updates = rho_updates; inf.reserved = 0;
        inf.obj_val = obj_val; inf.pri_res = pri_res; inf.dua_res = dua_res; inf.rho = rho;
        P.info[b] = inf; P.rho[b] = rho;
        atomicAdd(&P.stats[0], (unsigned long long)iter); atomicAdd(&P.stats[1], (unsigned long long)n_info);
        atomicAdd(&P.stats[2], (unsigned long long)rho_updates); atomicAdd(&P.stats[3], 1ULL);
    }
}

// ------------------------------------------------------------------------------------------------
// verification kernels
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(NT) void k_export(Lay L, Ptrs P, double *Pd, double *Ad_, double *q, double *l, double *u) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, p, S);
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
    double *p = sh; Smem S; smem_common(L, p, S);
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model};
    kkt_solve<NB>(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, rhs + (size_t)b * L.n, S.T, S.tv);
    for (int j = tid; j < L.n; j += NT) sol[(size_t)b * L.n + j] = S.T[j];
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
    L.NB = L.nb <= 4 ? 4 : (L.nb <= 8 ? 8 : (L.nb <= 16 ? 16 : 32));
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
    L.tsz = L.m > 4 * L.NB * L.NB ? L.m : 4 * L.NB * L.NB;
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
    if (Nc != Np) return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_create: control horizon Nc < Np is not implemented on the device yet");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPCQP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MPCQP_ERR_ARG, "mpcqp_create: bad device index");
    HIPCHK(hipSetDevice(device));
    mpcqp_handle *h = new mpcqp_handle();
    h->device = device; h->batch = batch; h->is_setup = false; h->u0_dev = nullptr;
    h->L = make_layout(nx, nu, Np, Nc);
    if (s) h->S = *s; else mpcqp_default_settings(&h->S);
    HIPCHK(hipStreamCreate(&h->stream));
    h->own_stream = true;
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
    rc |= dalloc(h, &P.dx, B * L.n); rc |= dalloc(h, &P.dy, B * L.m); rc |= dalloc(h, &P.rg, B * L.n);
    rc |= dalloc(h, &P.Dt, B * L.n); rc |= dalloc(h, &P.Et, B * L.m);
    rc |= dalloc(h, &P.ctype, B * L.m); rc |= dalloc(h, &P.info, B); rc |= dalloc(h, &P.stats, 8);
    rc |= dalloc(h, &h->u0_dev, B * L.nu);
    if (rc) { mpcqp_destroy(h); return MPCQP_ERR_HIP; }
    h->smem_setup = sizeof(double) * (size_t)smem_common_doubles(L);
    size_t with_state = h->smem_setup + sizeof(double) * (size_t)(L.n + 2 * L.m);
    h->lds_state = with_state <= 64 * 1024;      // keep >= 2 workgroups per CU; otherwise iterate in L2/HBM
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
    case 4:  { constexpr int NB = 4;  EXPR; } break; \
    case 8:  { constexpr int NB = 8;  EXPR; } break; \
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

static int launch_solve(mpcqp_handle *h, int plain_iters) {
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "solve before mpcqp_setup");
    HIPCHK(hipSetDevice(h->device));
    const Lay &L = h->L;
    DISPATCH_NB(L.NB, {
        if (h->lds_state) {
            if (set_smem(k_solve<NB, true>, h->smem_solve)) return MPCQP_ERR_HIP;
            hipLaunchKernelGGL((k_solve<NB, true>), dim3(h->batch), dim3(NT), h->smem_solve, h->stream, h->L, h->P, h->S, plain_iters);
        } else {
            if (set_smem(k_solve<NB, false>, h->smem_solve)) return MPCQP_ERR_HIP;
            hipLaunchKernelGGL((k_solve<NB, false>), dim3(h->batch), dim3(NT), h->smem_solve, h->stream, h->L, h->P, h->S, plain_iters);
        }
    });
    HIPCHK(hipGetLastError());
    return MPCQP_OK;
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
    if (reset) HIPCHK(hipMemsetAsync(h->P.stats, 0, 4 * sizeof(uint64_t), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
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
        int64_t full = (int64_t)(L.Nc) * (nb * (nb + 1) / 2) + (int64_t)(L.N - L.Nc) * (nx * (nx + 1) / 2);
        int64_t sub = (int64_t)(L.Nc - 1) * (nx * nb + (int64_t)L.nu * L.nu) + (int64_t)(L.N - L.Nc) * nx * nb;
        *nnzL = full + sub + 2 * (int64_t)L.n_x;
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

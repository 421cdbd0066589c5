// mpcqp_huge.h -- stage blocks wider than 64: 64 < nx + nu <= 128 (the reference accepts any size, mpc.py:82-105).
//
// A MERELY CORRECT backend, so that setup() does not refuse shapes the reference takes: the same one-directional block LDL' as the 64-wide
// stages (mpcqp_wide.h) --
//     S_0 = K_00,   M_k = K_{k+1,k} S_k^-1,   S_{k+1} = K_{k+1,k+1} - M_k K_{k+1,k}'
//     forward  y_{k+1} = r_{k+1} - M_k y_k;    z_k = S_k^-1 y_k;    backward  x_k = z_k - M_k' x_{k+1}
// -- with nothing tuned: the 128 x 128 work matrices of the factorization do not fit LDS next to the work vectors, so they live in a global
// workspace (Gauss-Jordan in place there, pivot row and column staged through LDS), the block products are plain triple loops, and a mat-vec
// of the solve is one wave per row at a time (lanes along the row: coalesced loads, a wave reduction per row).  Per stage the factor holds
// [ S^-1 | M | M' ], row major.  Everything around the solve -- matrix-free QP, parallel phases on the global-memory iterate, checks, the
// bordered correction for Nc < Np -- is the code every other width runs, instantiated at stride 128.
#pragma once

struct HugeFmt {
    static constexpr int NB = 128, NN = NB * NB;
    static constexpr int STAGE = 3 * NN, OSINV = 0, OM = NN, OMT = 2 * NN;
    static constexpr int WS = 2 * NB + 8;                     // LDS doubles of the factorization: pivot row and column
    static constexpr int GWS = 2 * NN;                        // global workspace per instance: S (evolving), C (coupling block)
};

// y = M v for one NB x NB row-major matrix in global memory, v in LDS: wave w takes rows w, w + 4, ...; f(r, sum) on lane 0
template <class Fn>
__device__ __forceinline__ void huge_matvec(const double *M, const double *v, Fn &&f) {
    constexpr int NB = HugeFmt::NB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    cgdouble *Mg = (cgdouble *)M;
    const double v0 = v[lane], v1 = v[lane + 64];
    for (int r = wv; r < NB; r += NWAVES) {
        double a = Mg[(size_t)r * NB + lane] * v0 + Mg[(size_t)r * NB + lane + 64] * v1;
        a = wave_reduce<false>(a);
        if (lane == 0) f(r, a);
    }
}

// Tc <- K^-1 Tc on the padded stage-major vector (stride 128); Zc: LDS, N * 128 doubles.  All threads call; barriers inside.
__device__ __forceinline__ void huge_core(const double *F, int N, double *Tc, double *Zc) {
    constexpr int NB = HugeFmt::NB;
    for (int k = 0; k + 1 < N; ++k) {
        huge_matvec(F + (size_t)k * HugeFmt::STAGE + HugeFmt::OM, Tc + k * NB, [&](int r, double t) { Tc[(k + 1) * NB + r] -= t; });
        __syncthreads();
    }
    for (int k = 0; k < N; ++k) huge_matvec(F + (size_t)k * HugeFmt::STAGE + HugeFmt::OSINV, Tc + k * NB, [&](int r, double t) { Zc[k * NB + r] = t; });
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += NT) Tc[(N - 1) * NB + i] = Zc[(N - 1) * NB + i];
    __syncthreads();
    for (int k = N - 2; k >= 0; --k) {
        huge_matvec(F + (size_t)k * HugeFmt::STAGE + HugeFmt::OMT, Tc + (k + 1) * NB, [&](int r, double t) { Tc[k * NB + r] = Zc[k * NB + r] - t; });
        __syncthreads();
    }
}

// Factorization.  Wg: this instance's global workspace (HugeFmt::GWS doubles); W: LDS, HugeFmt::WS doubles.  Returns 1 on a non-positive pivot.
__device__ __forceinline__ int factor_huge(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *Wg, double *W, int *iflag) {
    constexpr int NB = HugeFmt::NB, NN = HugeFmt::NN;
    const Lay &L = c.L;
    const int N = L.N, tid = threadIdx.x;
    double *S = Wg, *C = Wg + NN, *prow = W, *pcol = W + NB;
    if (tid == 0) *iflag = 0;
    for (int e = tid; e < NN; e += NT) S[e] = kkt_diag_entry<true>(c, om, sv, cc, 0, e / NB, e % NB);
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        double *Fk = F + (size_t)k * HugeFmt::STAGE;
        for (int pv = 0; pv < NB; ++pv) {                       // in-place Gauss-Jordan: every entry from its own old value and the OLD pivot row / column
            for (int i = tid; i < NB; i += NT) { prow[i] = S[pv * NB + i]; pcol[i] = S[i * NB + pv]; }
            __syncthreads();
            double d = prow[pv];
            if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
            const double inv = 1.0 / d;
            for (int e = tid; e < NN; e += NT) {
                const int i = e / NB, j = e % NB;
                const double t = pcol[i] * inv;
                S[e] = (i == pv) ? (j == pv ? inv : prow[j] * inv) : (j == pv ? -t : fma(-t, prow[j], S[e]));
            }
            __syncthreads();
        }
        for (int e = tid; e < NN; e += NT) { const int a = e / NB, b = e % NB; Fk[HugeFmt::OSINV + e] = 0.5 * (S[a * NB + b] + S[b * NB + a]); }      // the TRUE inverse for the products below
        __syncthreads();
        if (k + 1 < N) {
            for (int e = tid; e < NN; e += NT) C[e] = kkt_sub_entry(c, om, cc, k, e / NB, e % NB);      // K_{k+1,k}
            __syncthreads();
            for (int e = tid; e < NN; e += NT) {               // M = C S^-1
                const int a = e / NB, b = e % NB;
                double acc = 0.0;
                for (int l = 0; l < NB; ++l) acc = fma(C[a * NB + l], Fk[HugeFmt::OSINV + l * NB + b], acc);
                Fk[HugeFmt::OM + e] = acc; Fk[HugeFmt::OMT + b * NB + a] = acc;
            }
            __syncthreads();
            for (int e = tid; e < NN; e += NT) {               // S_{k+1} = K_{k+1,k+1} - M C'
                const int a = e / NB, b = e % NB;
                double acc = kkt_diag_entry<true>(c, om, sv, cc, k + 1, a, b);
                for (int l = 0; l < NB; ++l) acc = fma(-Fk[HugeFmt::OM + a * NB + l], C[b * NB + l], acc);
                S[e] = acc;
            }
            __syncthreads();
        }
        // the STORED inverse has zero rows and columns where the stage has no variable (padding, inputs beyond the control horizon), like everywhere else
        const int nbk = (k < L.NcT) ? L.nb : L.nx;
        for (int e = tid; e < NN; e += NT) if (e / NB >= nbk || e % NB >= nbk) Fk[HugeFmt::OSINV + e] = 0.0;
        __syncthreads();
    }
    return *iflag;
}

// the two entry points the kernels call, for 128-wide stages
template <>
__device__ __forceinline__ int factor_all<128>(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag, BorderPtrs bp) {
    const int bad = factor_huge(c, om, sv, cc, F, bp.gws, W, iflag);
    if (c.L.border) border_factor<128>(c, om, sv, cc, F, bp.Bb, bp.Zb, bp.Sig, W, W + c.L.m, bp.red);
    return bad;
}
template <>
__device__ __forceinline__ void kkt_core<128, false>(const CoreArgs &a, double *Tc) { huge_core(a.F, a.N, Tc, Tc + a.N * HugeFmt::NB); TICK(3) }

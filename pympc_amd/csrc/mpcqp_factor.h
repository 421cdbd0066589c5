// mpcqp_factor.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Twisted block LDL' factorization of the reduced KKT matrix into FP64-MFMA operand fragments.
#pragma once

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
// (Stage record formats: FactorFmt below.)
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
typedef gdouble gdouble_g;                                     // (for signatures of templates that re-typedef gdouble locally)
typedef __attribute__((address_space(1))) const d4 cgd4;

// The factor of one instance.  Per stage  [ forward matrix -Mh_k | packed S_k^-1 | table ]  and, behind the last stage, the
// constant fragments [ G | G' ],  G = [[Ad, Bd], [0, c QDu']]:
//   * forward elimination runs from the forward matrices (one MFMA mat-vec per stage, chain_sweep);
//   * there is NO separate S^-1 phase and no second read of the forward matrices: the back substitution is
//         x_k = S_k^-1 ( yh_k - K_{k,nbr} x_nbr )          (because Mh_{k+1}' = S_k^-1 K_{k,k+1})
//     with the off-diagonal block K_{k,k+-1} applied MATRIX-FREE from G / G' (held in registers for a whole sweep), scaled by
//     the stage's omega (so_sweep in mpcqp_sweeps.h);
//   * S_k^-1 is symmetric and stored packed: 16 x 16 stages  sym(S) = 164 doubles;  32 x 32 stages
//     [ sym(S00) 164 | sym(S11) 164 | S01 256 ]  (mpcqp_sweeps.h: sym_window / sym_expand16 / frag_transpose16), and with ZERO rows
//     and columns where the stage has no variable (padding, inputs beyond the control horizon) instead of the identity, so that
//     the back substitution needs no row mask;
//   * the table holds what the matrix-free product needs of omega: per stage element (scale, Delta-u coupling weight) towards
//     the neighbour the back substitution comes from, written by the factorization (omega changes only when it runs).
// Per stage and iteration the sweeps stream NB^2 + SINV + 2 NB doubles (16: 3.6 KB, 32: 13.4 KB) instead of the 2 NB^2 + SINV of
// a format that also runs the back substitution from (transposed) forward matrices.
// 32 x 32 stages use the S^-1-ONLY variant (MPCQP_SONLY32 = 1, the default): NO forward matrices -- header first, per stage
// [ packed S^-1 | table towards the stage above | table towards the stage below ] -- and both sweeps run through S^-1 and G:
//     forward   w_k = S_k^-1 ( b_k - K_{k,nbr} w_nbr )        backward   x_k = w_k - S_k^-1 K_{k,nbr} x_nbr
// 2 x 5.7 KB per stage and iteration instead of 8 + 4.7 + 0.5 KB.  Measured on cfg-5 (scripts/diag/ab_cfg5.sh): the forward
// matrices cost 12 % throughput at 1024 instances and 7 % at the parity tolerance (the stream is the bound there), and win
// 7 % on the 512-instance default run, where one straggling instance's critical path sets the time; MPCQP_SONLY32 = 0 builds
// that alternative.  16 x 16 stages: the forward-matrix format wins at every batch size (+7 .. 10 % over reading S^-1 twice).
#ifndef MPCQP_SONLY32
#define MPCQP_SONLY32 1
#endif
template <int NB> struct FactorFmt {
    static constexpr bool SONLY = MPCQP_SONLY32 && NB == 32;
    static constexpr int SINV = NB == 32 ? 164 + 164 + 256 : 164;
    static constexpr int FWD = SONLY ? 0 : NB * NB;
    static constexpr int TAB = SONLY ? 4 * NB : 2 * NB;
    static constexpr int STAGE = FWD + SINV + TAB;
    static constexpr int HEAD = 2 * NB * NB;                              // [G | G']: behind the stages (in front of them in the S^-1-only format)
    static constexpr int SOFF = FWD;                                    // offset of S^-1 inside a stage
};
typedef double d4u __attribute__((ext_vector_type(4), aligned(8)));
typedef __attribute__((address_space(1))) const d4u cgd4u;
__host__ __device__ inline int sym_cum(int R) { return 16 * R - 2 * R * (R - 1); }
__device__ __forceinline__ int sym_pos(int r, int c) {       // (r, c) with c>>2 >= r>>2
    const int R = r >> 2;
    return 40 * (c & 3) + sym_cum(R) + (r & 3) * (4 - R) + ((c >> 2) - R);
}
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
struct BorderPtrs { double *Bb, *Zb, *Sig, *red; double *gws; };      // (gws: the global factorization workspace of 128-wide stages, mpcqp_huge.h)

template <int NB> __device__ __forceinline__ void border_factor(const Ctx &, const double *, const double *, double, const double *, double *, double *, double *, double *, double *, double *);

// The two half-chains are independent until the middle stage: for 16 x 16 stages each is factored by one half of the
// workgroup, side by side (the stage count, not the arithmetic, is what a refactorization costs).  32 x 32 stages keep the
// whole workgroup on one stage at a time (four entries per thread; two workspaces would not fit next to a second
// resident workgroup).  S_k^-1 comes from an in-place Gauss-Jordan sweep of S_k -- NB steps, every entry updated in
// parallel -- instead of Cholesky + a thread-serial triangular inverse + L^-T L^-1; S_k is positive definite, its pivots
// are the Schur complements a Cholesky would take roots of, and a non-positive one is reported the same way.
template <int NB> struct FactorCfg {
    static constexpr int G = NB == 16 ? 2 : 1;          // thread groups working on different stages
    static constexpr int WS = NB == 16 ? 8 * NB * NB : 5 * NB * NB;      // LDS doubles: per group [S | Ks | Mh | Sn], or [S | Ks | Mh | SnA | SnB]
};

template <int NB>
__device__ __forceinline__ int factor_all(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag, BorderPtrs bp) {
    const Lay &L = c.L;
    constexpr int G = FactorCfg<NB>::G, NN = NB * NB;
    const int tid = threadIdx.x;
    const int N = L.N, mid = N / 2;
    if (tid == 0) *iflag = 0;
    TICK_RESET
    double *SnA, *SnB;                                   // S^-1 of the stage eliminated last in the top / bottom half
    if (G == 2) { SnA = W + 3 * NN; SnB = W + 7 * NN; } else { SnA = W + 3 * NN; SnB = W + 4 * NN; }
    // One stage by the T threads lt = 0..T-1 of a group (barriers are workgroup-wide: every thread makes the same calls;
    // `on` = this group has a stage in this call).  Ws: the group's [S | Ks | Mh]; SnU / SnD: S^-1 of the neighbours above /
    // below to eliminate (null = none); SnOut receives S_k^-1.
    auto stage = [&](bool on, int k, int T, int lt, double *Ws, const double *SnU, const double *SnD, double *SnOut) {
        double *S = Ws, *Ks = Ws + NN, *Mh = Ws + 2 * NN;
        // K_kk.  Its one dense term, G' diag(om_dyn) G with G = [Ad Bd] (nx rows) and om_dyn the weights of the dynamics rows of stage k+1, is
        // formed as a product from LDS, with the same control flow for every entry: G zero-padded to NB columns in Ks, om_dyn in Mh (both idle
        // here).  Entry by entry (kkt_diag_entry<true>) it is three divergent nx-long loops and a global fetch per term.
        const int nbk = (k < L.NcT) ? L.nb : L.nx;
        const bool dyn = k < L.Np;
        if (on && dyn) {
            for (int e = lt; e < L.nx * NB; e += T) {
                const int r = e / NB, q = e % NB;
                Ks[e] = q < L.nx ? c.Ad()[r * L.nx + q] : (q < nbk ? c.Bd()[r * L.nu + (q - L.nx)] : 0.0);
            }
            if (lt < L.nx) Mh[lt] = om[(k + 1) * L.nx + lt];
        }
        __syncthreads();
        TICK_START
        if (on) for (int e = lt; e < NN; e += T) {
            const int a = e / NB, b = e % NB;
            double v = kkt_diag_entry<false>(c, om, sv, cc, k, a, b);
            if (dyn) {
#pragma unroll 4
                for (int r = 0; r < L.nx; ++r) v = fma(Ks[r * NB + a] * Mh[r], Ks[r * NB + b], v);
            }
            S[e] = v;
            if constexpr (!FactorFmt<NB>::SONLY) { if (k == N - 1) F[(size_t)k * L.fstage + e] = 0.0; }     // the last stage has no forward matrix (stage 0's slot holds the middle's second one)
        }
        for (int side = 0; side < 2; ++side) {           // S -= (Ks Sn) Ks' for the neighbour above (side 0) / below (side 1)
            const double *Sn = side == 0 ? SnU : SnD;
            const bool has = on && Sn != nullptr;
            const bool up = side == 0;
            __syncthreads();
            TICK(0)
            // The two block products run on the matrix cores (v_mfma_f64_16x16x4_f64: one 16 x 16 tile of C += A B per wave and instruction, four
            // steps of K per instruction; lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15] and receives C[(l >> 4) + 4 v][l & 15], v = 0..3).
            // Both operands are read along their fast index: the coupling block is stored transposed (KsT[b][a] = K_{k,nbr}[a][b]), Sn is
            // symmetric, and the first product leaves Mh transposed for the second.  One tile per wave: wave 0 of a 16 x 16 group, all four
            // waves at 32 x 32.
            if (has) for (int e = lt; e < NN; e += T) {
                const int a = e / NB, b = e % NB;              // KsT[a][b] = Ks[b][a]
                Ks[e] = up ? kkt_sub_entry(c, om, cc, k - 1, b, a) : kkt_sub_entry(c, om, cc, k, a, b);
            }
            __syncthreads();
            TICK(1)
            constexpr int TW = NB / 16;
            const int wg = lt >> 6, ln = lt & 63, ti = wg / TW, tj = wg % TW, lr = ln & 15, lk = ln >> 4;
            const bool mm = has && wg < TW * TW;                 // (uniform per wave)
            const int fwd_stage = (k == mid && !up) ? 0 : k;
            if (mm) {                                            // Mh = Ks Sn
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < NB / 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ks[(4 * kk + lk) * NB + 16 * ti + lr], Sn[(4 * kk + lk) * NB + 16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v, col = 16 * tj + lr;
                    Mh[col * NB + row] = acc[v];                 // (transposed: the second product reads it as its A operand)
                    // forward matrix of stage k; the middle stage's second forward matrix lives in the otherwise unused slot of stage 0
                    if constexpr (!FactorFmt<NB>::SONLY) F[(size_t)fwd_stage * L.fstage + frag_pos<NB>(row, col)] = -acc[v];
                }
            }
            __syncthreads();
            TICK(2)
            if (mm) {                                            // S -= Mh Ks'
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < NB / 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Mh[(4 * kk + lk) * NB + 16 * ti + lr], Ks[(4 * kk + lk) * NB + 16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) S[(16 * ti + lk + 4 * v) * NB + 16 * tj + lr] -= acc[v];
            }
        }
        __syncthreads();
        TICK(3)
        // in-place Gauss-Jordan inversion of the SPD block: step p uses the OLD pivot row and column (read, barrier, write)
        constexpr int EPT = NN / (NT / G);                   // entries per thread (2 with two groups of 128, 4 for 32 x 32), kept in registers through the NB steps
        static_assert(kLatOnly || NN % (NT / G) == 0, "whole entries per thread");
        double cur[EPT];
        if (on) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) cur[u] = S[lt + u * T];
        }
        static_assert(NB % 2 == 0, "the pivot steps alternate between two buffers and end in S");
        for (int pv = 0; pv < NB; ++pv) {
            // (one barrier per step: step pv reads the matrix step pv - 1 wrote and writes the other buffer -- Mh is idle here -- so nobody
            //  overwrites what a slower thread of the same step is still reading)
            const double *Sr = (pv & 1) ? Mh : S;
            double *Sw = (pv & 1) ? S : Mh;
            double rip[EPT], rpj[EPT], d = 1.0;
            if (on) {
                d = Sr[pv * NB + pv];
#pragma unroll
                for (int u = 0; u < EPT; ++u) { const int e = lt + u * T; rip[u] = Sr[(e / NB) * NB + pv]; rpj[u] = Sr[pv * NB + (e % NB)]; }
            }
            if (on && !(d > 0.0)) { if (lt == 0) *iflag = 1; d = 1e-300; }
            if (on) {
                const double inv = 1.0 / d;
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int e = lt + u * T, i = e / NB, j = e % NB;
                    const bool rowp = i == pv, colp = j == pv;
                    const double t = rip[u] * inv;
                    const double off = rowp ? rpj[u] * inv : fma(-t, rpj[u], cur[u]), onp = rowp ? inv : -t;
                    cur[u] = colp ? onp : off;
                    Sw[e] = cur[u];
                }
            }
            __syncthreads();
        }
        TICK(4)
        if (on) for (int e = lt; e < NN; e += T) {               // S now holds S_k^-1: symmetrise, keep, store in the factor's format
            const int a = e / NB, b = e % NB;
            const double acc = 0.5 * (S[a * NB + b] + S[b * NB + a]);
            SnOut[e] = acc;
            // (rows / columns without a variable are stored as zero -- see FactorFmt; the LDS copy SnOut stays the true inverse)
            const bool dead = a >= L.nb || b >= L.nb || ((a >= L.nx || b >= L.nx) && k >= L.NcT);
            const double st = dead ? 0.0 : acc;
            double *Sk = F + (FactorFmt<NB>::SONLY ? FactorFmt<NB>::HEAD : 0) + (size_t)k * L.fstage + FactorFmt<NB>::SOFF;
            if constexpr (NB == 32) {                              // [ sym(S00) | sym(S11) | S01 ]
                const int A_ = a >> 4, B_ = b >> 4, al = a & 15, bl = b & 15;
                if (A_ == B_) { if ((bl >> 2) >= (al >> 2)) Sk[164 * A_ + sym_pos(al, bl)] = st; }
                else if (A_ == 0) Sk[328 + frag_pos<16>(al, bl)] = st;
            } else if ((b >> 2) >= (a >> 2)) Sk[sym_pos(a, b)] = st;
        }
        {
            // tables of the stage's off-diagonal blocks (kkt_sub_entry is the entry-wise definition): element e -> (omega of the
            // dynamics row between the two stages, 1 off the x part;  omega of the Delta-u row that couples them, on the one
            // element of this stage it enters, 0 elsewhere)
            if (on && lt < NB) {
                const int e = lt;
                auto entry = [&](int nbr, double *dst) {
                    double sc = 1.0, cw = 0.0;
                    if (nbr >= 0 && nbr < N) {
                        const int hi = max(k, nbr), lo = min(k, nbr);
                        if (e < L.nx) sc = om[hi * L.nx + e];
                        const int edst = nbr > k ? L.nx + L.nu - 1 : L.nx;
                        if (e == edst && hi < L.NcT) cw = om[L.rdu + L.nu + min(lo, max(L.NcT - 2, 0)) * L.nu + L.nu - 1];
                    }
                    dst[2 * e] = sc; dst[2 * e + 1] = cw;
                };
                double *tab = F + (FactorFmt<NB>::SONLY ? FactorFmt<NB>::HEAD : 0) + (size_t)k * L.fstage + FactorFmt<NB>::FWD + FactorFmt<NB>::SINV;
                if constexpr (FactorFmt<NB>::SONLY) { entry(k - 1, tab); entry(k + 1, tab + 2 * NB); }
                else entry(k == mid ? -1 : (k < mid ? k + 1 : k - 1), tab);
            }
        }
        TICK(5)
    };
    if (G == 2) {
        const int T = NT / 2, g = tid / T, lt = tid % T;
        double *Ws = W + g * 4 * NN, *Sn = g == 0 ? SnA : SnB;
        const int nA = mid, nB = N - 1 - mid, steps = nA > nB ? nA : nB;
        for (int t = 0; t < steps; ++t) {
            const bool on = g == 0 ? t < nA : t < nB;
            const int k = g == 0 ? t : N - 1 - t;
            // (a group's Sn is its own previous output: reading it as the neighbour and overwriting it at the end of the
            //  same call are separated by the barriers of the inversion)
            stage(on, k, T, lt, Ws, (g == 0 && t > 0) ? Sn : nullptr, (g == 1 && t > 0) ? Sn : nullptr, Sn);
        }
    } else {
        for (int k = 0; k < mid; ++k) stage(true, k, NT, tid, W, k > 0 ? SnA : nullptr, nullptr, SnA);
        for (int k = N - 1; k > mid; --k) stage(true, k, NT, tid, W, nullptr, k < N - 1 ? SnB : nullptr, SnB);
    }
    stage(true, mid, NT, tid, W, SnA, SnB, SnA);
    {   // G = [[Ad, Bd], [0, c QDu']] and G' as fragments (constant per instance)
        constexpr int NBLK = NB / 16;
        double *Fg = FactorFmt<NB>::SONLY ? F : F + (size_t)N * L.fstage;
        for (int e = tid; e < NB * NB; e += NT) {
            const int r = e / NB, q = e % NB;
            double g = 0.0;
            if (r < L.nx) g = q < L.nx ? c.Ad()[r * L.nx + q] : (q < L.nb ? c.Bd()[r * L.nu + (q - L.nx)] : 0.0);
            else if (r < L.nb && q >= L.nx && q < L.nb) g = cc * c.QDu()[(q - L.nx) * L.nu + (r - L.nx)];
            Fg[frag_pos<NB>(r, q)] = g;
            Fg[NB * NB + frag_pos<NB>(q, r)] = g;
        }
        (void)NBLK;
    }
    TICK_FLUSH
    __syncthreads();
    if (L.border) border_factor<NB>(c, om, sv, cc, F, bp.Bb, bp.Zb, bp.Sig, W, W + L.m, bp.red);
    return *iflag;
}
template <> __device__ __forceinline__ int factor_all<64>(const Ctx &, const double *, const double *, double, double *, double *, int *, BorderPtrs);      // mpcqp_wide.h
template <> __device__ __forceinline__ int factor_all<128>(const Ctx &, const double *, const double *, double, double *, double *, int *, BorderPtrs);     // mpcqp_huge.h

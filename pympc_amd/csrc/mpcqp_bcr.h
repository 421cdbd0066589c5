// mpcqp_bcr.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Block cyclic reduction of the block-tridiagonal reduced KKT matrix: the latency backend for 16 x 16 stages.
//
// The twisted block LDL' (mpcqp_factor.h / mpcqp_sweeps.h) is a chain: N/2 dependent stage steps forward, N/2 backward, two of
// the four waves busy -- 5.5 of the 9.7 us an ADMM iteration of ONE (12,4,30) instance takes on an otherwise idle CU.  Odd-even
// (cyclic) reduction is the same block Gaussian elimination in nested-dissection order: level l eliminates every other stage of
// what is left (stride h = 2^l), so the dependent depth is 2 ceil(log2 N) level steps instead of N stage steps, every level
// step is a set of INDEPENDENT 16 x 16 mat-vecs, and all four waves work.  Elimination order changes nothing about
// stability: the pivots are Schur complements of a symmetric positive definite matrix in either order.
//
//   level l, eliminated stages E_l = { e = h (2t+1) - 1 },  kept stages A_{l+1} = { i = 2h (t+1) - 1 },  neighbours e -+ h:
//     factor   D_e^-1 = inv(K_ee);  Lb_{e,i} = D_e^-1 K_{e,i}  (i = e-h, e+h);
//              K_ii -= K_{i,e} Lb_{e,i};   K_{e-h,e+h} = -K_{e-h,e} Lb_{e,e+h}          (the reduced chain of the next level)
//     forward  c_e = D_e^-1 b_e;   b_i -= Lb_{e,i}' b_e            (kept i pulls from its two eliminated neighbours)
//     backward x_e = c_e - Lb_{e,e-h} x_{e-h} - Lb_{e,e+h} x_{e+h}
// Per stage the factor holds five MFMA fragments (operand order of v_mfma_f64_4x4x4_4b_f64, mpcqp_factor.h):
//     [ D^-1 | -Lb_L | -Lb_R | -Lb_L' | -Lb_R' ]        (L / R: towards the neighbour e-h / e+h)
// 31 stages: 143 fragments used per solve = 286 KB -- 2.5 x the bytes of the chain format, which is why this backend is chosen
// for batches of at most one instance per compute unit: there the FOUR WAVES' REGISTER FILES hold the whole factor (each wave
// the fragments of the tasks it always executes: 43 slots x 8 VGPRs of the 512 a wave may have at one workgroup per CU), loaded
// once per round of check_termination iterations -- an iteration then reads no factor at all (mpcqp_lat.h).
#pragma once
#include <type_traits>

#include "mpcqp_topv.h"

struct BcrFmt {
    static constexpr int NN = 256;                            // doubles per 16 x 16 fragment
    static constexpr int REC = 5 * NN;                        // [ D^-1 | -LbL | -LbR | -LbL' | -LbR' ]
    static constexpr int ODINV = 0, OLBL = NN, OLBR = 2 * NN, OLBLT = 3 * NN, OLBRT = 4 * NN;
    static constexpr int WSTAGE = 4 * NN;                     // global workspace per stage of the factorization: K_ii | K_{i,next} | dK_L | dK_R
    // LDS of the factorization: per wave [ D | B_L | B_R | Lb_L | Lb_R ] for the levels; omega and s during the assembly; two block rows of the top inverse + two blocks
    static constexpr int lds_doubles(int waves, int m_plus_n, int nt) { const int a = waves * 5 * NN, b = (2 * nt + 2) * NN; return (a > b ? a : b) > m_plus_n ? (a > b ? a : b) : m_plus_n; }
    // Dense top (Lay::bcrtop = nt > 0): the reduction stops after levels 0 and 1; the nt = N / 4 stages i = 4 (r + 1) - 1 that are left form a block
    // tridiagonal Schur complement of order 16 nt (112 at N = 31) whose EXPLICIT inverse is stored as nt x nt fragments behind the stage records:
    // block (r, c) at TOPOFF(N) + (r nt + c) NN.  Levels 2 .. 4 of the plain reduction are a chain of five dependent level steps that one wave
    // walks alone (2 700 of an iteration's 11 000 cycles); the inverse is nt independent block rows of nt mat-vecs each.
    // LATW_TOP_VALU (mpcqp_latw.h: the round applies the inverse on the vector ALU): a second copy of it behind the fragments, in the row-part order the
    // round keeps it in LDS in -- TOPVOFF(N) + 2 pair + {0, 1}, pairs as bcr_topv_rc enumerates them -- so that the round's LDS copy is a straight, coalesced copy.
    static constexpr int top_count(int N) { return N / 4; }
    static constexpr long long top_off(int N) { return (long long)N * REC; }
    static constexpr long long topv_off(int N) { return top_off(N) + (long long)top_count(N) * top_count(N) * NN; }
    static constexpr long long doubles(int N, bool top) { return (long long)N * REC + (top ? (LATW_TOP_VALU ? 2LL : 1LL) * top_count(N) * top_count(N) * NN : 0); }
};

// ------------------------------------------------------------------------------------------------
// Factorization (run time N).  Wg: this instance's global workspace, N * WSTAGE doubles; W: LDS, BcrFmt::lds_doubles(waves, m + n).
// Every 16 x 16 block operation belongs to ONE wave (lane (a0 = lane >> 4, b = lane & 15) holds the elements (a0 + 4u, b), u < 4, of a block):
// a wave eliminates a stage of a level on its own -- inversion, the two products with the couplings, the Schur contributions -- in its own five
// LDS blocks, ordered by the wave's in-order LDS pipe (wave_sync: no workgroup barrier inside a task; round 5's version ran four stages side by
// side behind 38 workgroup barriers each and cost 130 us per instance).  Workgroup barriers separate the levels.  Returns 1 on a non-positive pivot.
// N = L.bcr >= L.N is the stage count of the elimination tree (the register-resident schedule of mpcqp_lat.h exists for a few sizes): the
// stages beyond the problem's own are identity blocks without couplings -- they cost the schedule a few idle mat-vecs and change nothing.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// in-place Gauss-Jordan inversion of an SPD 16 x 16 block in LDS by one wave (step p uses the OLD pivot row and column), then symmetrised
__device__ __forceinline__ int wave_inv16(double *D, int lane) {
    const int b = lane & 15, a0 = lane >> 4;
    double v[4];
    int bad = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = D[(a0 + 4 * u) * 16 + b];
#pragma unroll 2
    for (int pv = 0; pv < 16; ++pv) {
        double d = D[pv * 17];
        const double rpj = D[pv * 16 + b];
        double rip[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rip[u] = D[(a0 + 4 * u) * 16 + pv];
        if (!(d > 0.0)) { bad = 1; d = 1e-300; }
        const double inv = 1.0 / d;
        wave_sync();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = a0 + 4 * u;
            v[u] = (i == pv) ? (b == pv ? inv : rpj * inv) : (b == pv ? -rip[u] * inv : v[u] - rip[u] * rpj * inv);
            D[i * 16 + b] = v[u];
        }
        wave_sync();
    }
    double sy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sy[u] = 0.5 * (v[u] + D[b * 16 + a0 + 4 * u]);
    wave_sync();
#pragma unroll
    for (int u = 0; u < 4; ++u) D[(a0 + 4 * u) * 16 + b] = sy[u];
    wave_sync();
    return bad;
}
// o(a, b) += sum_l X(a, l) Y(l, b), X read as X[a][l] or (XT) as X[l][a]: broadcast reads of X, row reads of Y -- no bank conflicts
template <bool XT>
__device__ __forceinline__ void wave_mul16(const double *X, const double *Y, int lane, double (&o)[4]) {
    const int b = lane & 15, a0 = lane >> 4;
#pragma unroll 8
    for (int l = 0; l < 16; ++l) {
        const double y = Y[l * 16 + b];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = fma(XT ? X[l * 16 + a0 + 4 * u] : X[(a0 + 4 * u) * 16 + l], y, o[u]);
    }
}

__device__ __forceinline__ int factor_bcr(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *Wg, double *W, int *iflag) {
    constexpr int NN = BcrFmt::NN, G = NT / 64, NB = 16;
    const Lay &L = c.L;
    const int N = L.bcr, NR = L.N, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, b = lane & 15, a0 = lane >> 4;
    double *Kd = Wg, *Up = Wg + (size_t)N * NN, *dKL = Up + (size_t)N * NN, *dKR = dKL + (size_t)N * NN;
    if (tid == 0) *iflag = 0;
    // (omega and s through an LDS copy at the start of the workspace -- every entry of a diagonal block sums nx products of them; the levels' blocks take its place)
    double *oml = W, *svl = oml + L.m;
    for (int r = tid; r < L.m; r += NT) oml[r] = om[r];
    for (int r = tid; r < L.n; r += NT) svl[r] = sv[r];
    __syncthreads();
    for (int idx = tid; idx < N * NN; idx += NT) {
        const int k = idx / NN, r = idx % NN, a = r / NB, bb = r % NB;
        Kd[idx] = k < NR ? kkt_diag_entry(c, oml, svl, cc, k, a, bb) : (a == bb ? 1.0 : 0.0);
        Up[idx] = (k + 1 < NR) ? kkt_sub_entry(c, oml, cc, k, bb, a) : 0.0;     // K_{k,k+1}[a][b] = K_{k+1,k}[b][a]
    }
    __syncthreads();
    int bad = 0;
    double *D = W + wv * 5 * NN, *BL = D + NN, *BR = D + 2 * NN, *LL = D + 3 * NN, *LR = D + 4 * NN;
    auto nvar = [&](int e) { return e >= NR ? 0 : (e < L.NcT) ? L.nb : L.nx; };      // variables of a stage; the rest is padding (identity in K, zero in the factor)
    for (int h = 1; h - 1 < N; h <<= 1) {
        if (L.bcrtop && h == 4) break;                        // (dense top: what is left after levels 0 and 1 is inverted explicitly below)
        const int ne = (N + h) / (2 * h);                     // stages e = h (2t+1) - 1 < N
        for (int t = wv; t < ne; t += G) {
            const int e = h * (2 * t + 1) - 1, i1 = e - h, i2 = e + h;
            const bool hasL = i1 >= 0, hasR = i2 < N;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + 4 * u, idx = a * NB + b;
                D[idx] = Kd[(size_t)e * NN + idx];
                BL[idx] = hasL ? Up[(size_t)i1 * NN + b * NB + a] : 0.0;     // K_{e,i1} = K_{i1,e}'
                BR[idx] = hasR ? Up[(size_t)e * NN + idx] : 0.0;            // K_{e,i2}
            }
            wave_sync();
            bad |= wave_inv16(D, lane);
            double accL[4] = {0.0, 0.0, 0.0, 0.0}, accR[4] = {0.0, 0.0, 0.0, 0.0};      // Lb = D^-1 B
            wave_mul16<false>(D, BL, lane, accL);
            wave_mul16<false>(D, BR, lane, accR);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int idx = (a0 + 4 * u) * NB + b; LL[idx] = accL[u]; LR[idx] = accR[u]; }
            wave_sync();
            double *rec = F + (size_t)e * BcrFmt::REC;
            const int nbk = nvar(e);
            double sL[4] = {0.0, 0.0, 0.0, 0.0}, sR[4] = {0.0, 0.0, 0.0, 0.0}, sU[4] = {0.0, 0.0, 0.0, 0.0};      // K_{i1,e} Lb_L,  K_{i2,e} Lb_R,  K_{i1,e} Lb_R
            wave_mul16<true>(BL, LL, lane, sL);
            wave_mul16<true>(BR, LR, lane, sR);
            if (hasL && hasR) wave_mul16<true>(BL, LR, lane, sU);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + 4 * u, idx = a * NB + b, fp = frag_pos<NB>(a, b);
                rec[BcrFmt::ODINV + fp] = (a >= nbk || b >= nbk) ? 0.0 : D[idx];
                rec[BcrFmt::OLBL + fp] = -accL[u];
                rec[BcrFmt::OLBR + fp] = -accR[u];
                rec[BcrFmt::OLBLT + fp] = -LL[b * NB + a];
                rec[BcrFmt::OLBRT + fp] = -LR[b * NB + a];
                dKL[(size_t)e * NN + idx] = sL[u];
                dKR[(size_t)e * NN + idx] = sR[u];
                if (hasL && hasR) Up[(size_t)i1 * NN + idx] = -sU[u];          // the coupling i1 <-> i2 of the next level
            }
            wave_sync();                                      // (the next task of this wave overwrites the blocks)
        }
        __syncthreads();
        // kept stages: Schur updates from the (one or two) eliminated neighbours, in a fixed order
        for (int idx = tid;; idx += NT) {
            const int t = idx / NN, r = idx % NN, i = 2 * h * (t + 1) - 1;
            if (i >= N) break;
            double v = Kd[(size_t)i * NN + r] - dKR[(size_t)(i - h) * NN + r];
            if (i + h < N) v -= dKL[(size_t)(i + h) * NN + r];
            Kd[(size_t)i * NN + r] = v;
        }
        __syncthreads();
    }
    if (L.bcrtop) {
        // ---- the top: stages i_r = 4 (r + 1) - 1, r < nt, with the Schur complements the two levels left in Kd (diagonal blocks A_r) and Up (Up[i_r] = B_r =
        // K_{i_r, i_{r+1}}): a block tridiagonal SPD matrix of order 16 nt whose EXPLICIT inverse the round applies.  Block LDL' down the chain,
        //     S_0 = A_0,   L_r = B_r' S_r^-1,   S_{r+1} = A_{r+1} - L_r B_r                     (one wave; S_r^-1 and L_r parked in the dead workspace dKL / dKR)
        // then the inverse from its last block row upwards (Sigma = D^-1 L^-1 + (I - L') Sigma, upper triangle):
        //     Sigma_{nt-1,nt-1} = S_{nt-1}^-1,   Sigma_{r,j} = -L_r' Sigma_{r+1,j} (j > r),   Sigma_{r,r} = S_r^-1 - Sigma_{r,r+1} L_r
        // with two block rows in LDS at a time (16 blocks = 32 KB; round 5 inverted the assembled 112 x 112 matrix by Gauss-Jordan sweeps in 103 KB of LDS --
        // one workgroup per compute unit, 224 barriers, seven times the arithmetic: 660 us per instance).  Every finished row goes out in both formats,
        // mirrored into the lower triangle, with zero rows and columns where a stage has no variable (as D^-1 above).
        const int nt = L.bcrtop;
        double *SinvG = dKL, *LG = dKR;
        if (wv == 0) {
            double *Sb = W, *Bb = W + NN, *Lb = W + 2 * NN;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int idx = (a0 + 4 * u) * NB + b; Sb[idx] = Kd[(size_t)3 * NN + idx]; }
            wave_sync();
            for (int r = 0; r < nt; ++r) {
                const int i = 4 * (r + 1) - 1;
                bad |= wave_inv16(Sb, lane);
                double sn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = (a0 + 4 * u) * NB + b;
                    SinvG[(size_t)r * NN + idx] = Sb[idx];
                    if (r + 1 < nt) { Bb[idx] = Up[(size_t)i * NN + idx]; sn[u] = Kd[(size_t)(i + 4) * NN + idx]; }
                }
                if (r + 1 == nt) break;
                wave_sync();
                double lacc[4] = {0.0, 0.0, 0.0, 0.0};
                wave_mul16<true>(Bb, Sb, lane, lacc);
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int idx = (a0 + 4 * u) * NB + b; Lb[idx] = lacc[u]; LG[(size_t)r * NN + idx] = lacc[u]; }
                wave_sync();
                double prod[4] = {0.0, 0.0, 0.0, 0.0};
                wave_mul16<false>(Lb, Bb, lane, prod);
                wave_sync();
#pragma unroll
                for (int u = 0; u < 4; ++u) Sb[(a0 + 4 * u) * NB + b] = sn[u] - prod[u];
                wave_sync();
            }
        }
        __syncthreads();
        double *rowP = W, *rowC = W + (size_t)nt * NN, *Lc = W + (size_t)2 * nt * NN, *Sc = Lc + NN;
        double *Ft = F + BcrFmt::top_off(N);
#if LATW_TOP_VALU
        double *Fv = F + BcrFmt::topv_off(N);                 // the same in the round's row-part order
#endif
        for (int idx = tid; idx < NN; idx += NT) rowC[(size_t)(nt - 1) * NN + idx] = SinvG[(size_t)(nt - 1) * NN + idx];
        for (int r = nt - 1; r >= 0; --r) {
            if (r < nt - 1) {
                for (int idx = tid; idx < NN; idx += NT) { Lc[idx] = LG[(size_t)r * NN + idx]; Sc[idx] = SinvG[(size_t)r * NN + idx]; }
                __syncthreads();
                for (int j = r + 1 + wv; j < nt; j += G) {
                    double o[4] = {0.0, 0.0, 0.0, 0.0};
                    wave_mul16<true>(Lc, rowP + (size_t)j * NN, lane, o);
#pragma unroll
                    for (int u = 0; u < 4; ++u) rowC[(size_t)j * NN + (a0 + 4 * u) * NB + b] = -o[u];
                    if (j == r + 1) {                         // ... and the diagonal block behind its right-hand neighbour, by the wave that has just written that one
                        wave_sync();
                        double dg[4] = {0.0, 0.0, 0.0, 0.0};
                        wave_mul16<false>(rowC + (size_t)j * NN, Lc, lane, dg);
                        double *Cd = rowC + (size_t)r * NN;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int idx = (a0 + 4 * u) * NB + b; dg[u] = Sc[idx] - dg[u]; Cd[idx] = dg[u]; }
                        wave_sync();
#pragma unroll
                        for (int u = 0; u < 4; ++u) dg[u] = 0.5 * (dg[u] + Cd[b * NB + a0 + 4 * u]);
                        wave_sync();
#pragma unroll
                        for (int u = 0; u < 4; ++u) Cd[(a0 + 4 * u) * NB + b] = dg[u];
                    }
                }
            }
            __syncthreads();
            const int nbr = nvar(4 * (r + 1) - 1);
            for (int idx = tid; idx < (nt - r) * NN; idx += NT) {
                const int j = r + idx / NN, e = idx % NN, a = e / NB, bb = e % NB;
                const double v = (a >= nbr || bb >= nvar(4 * (j + 1) - 1)) ? 0.0 : rowC[(size_t)j * NN + e];
                Ft[(size_t)(r * nt + j) * NN + frag_pos<NB>(a, bb)] = v;
                if (j != r) Ft[(size_t)(j * nt + r) * NN + frag_pos<NB>(bb, a)] = v;
#if LATW_TOP_VALU
                Fv[bcr_topv_pos(nt, NB * r + a, NB * j + bb)] = v;
                if (j != r) Fv[bcr_topv_pos(nt, NB * j + bb, NB * r + a)] = v;
#endif
            }
            __syncthreads();
            double *sw = rowP; rowP = rowC; rowC = sw;
        }
        if (tid == 0) iflag[2] = 0;                          // (the rounds' LDS copy of the top is stale -- and this workspace has just run over it: admm_latw)
    }
    if (bad) *iflag = 1;
    __syncthreads();
    return *iflag;
}

// one accumulating 16 x 16 mat-vec on the matrix cores: (p, q) += A in, as two dependent MFMA pairs (frag_matvec)
__device__ __forceinline__ void bcr_mv(const d4 a, double in, double &p, double &q) {
    const double r1 = rot_blocks<1>(in), r2 = rot_blocks<2>(in), r3 = rot_blocks<3>(in);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], in, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, q, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, q, 0, 0, 0);
}
__device__ __forceinline__ d4 bcr_frag(const double *F, int stage, int off, int lane) {
    return *(cgd4 *)(F + (size_t)stage * BcrFmt::REC + off + lane * 4);
}

// ------------------------------------------------------------------------------------------------
// Solve, streaming version (run time N; fragments read from memory as they are needed): the verification kernel and any caller
// outside an ADMM round.  Tc <- K^-1 Tc, Cc: LDS, N * 16 doubles.  All threads call; barriers inside.
// ------------------------------------------------------------------------------------------------
// ntop > 0: the factor has a dense top (BcrFmt): two levels of reduction, the explicit inverse for the ntop stages that are left, two levels back.
__device__ __forceinline__ void bcr_core_stream(const double *F, int N, double *Tc, double *Cc, int ntop = 0) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *tb = Tc + vec_lane_offset(lane), *cb = Cc + vec_lane_offset(lane);
    int top = 1;
    for (int h = 1; h - 1 < N; h <<= 1) {
        if (ntop && h == 4) break;
        top = h;
        const int ne = (N + h) / (2 * h), nk = N / (2 * h);
        for (int t = wv; t < nk; t += NWAVES) {
            const int i = 2 * h * (t + 1) - 1;
            double p = tb[i * 16], q = 0.0;
            bcr_mv(bcr_frag(F, i - h, BcrFmt::OLBRT, lane), tb[(i - h) * 16], p, q);
            if (i + h < N) bcr_mv(bcr_frag(F, i + h, BcrFmt::OLBLT, lane), tb[(i + h) * 16], p, q);
            tb[i * 16] = p + q;
        }
        for (int t = wv; t < ne; t += NWAVES) {
            const int e = h * (2 * t + 1) - 1;
            double p = 0.0, q = 0.0;
            bcr_mv(bcr_frag(F, e, BcrFmt::ODINV, lane), tb[e * 16], p, q);
            cb[e * 16] = p + q;
        }
        __syncthreads();
    }
    if (ntop) {                                               // x_top = (Schur complement)^-1 b_top, one block row per wave at a time
        const double *Ft = F + BcrFmt::top_off(N);
        for (int r = wv; r < ntop; r += NWAVES) {
            double p = 0.0, q = 0.0;
            for (int c = 0; c < ntop; ++c) bcr_mv(*(cgd4 *)(Ft + (size_t)(r * ntop + c) * BcrFmt::NN + lane * 4), tb[(4 * (c + 1) - 1) * 16], p, q);
            cb[(4 * (r + 1) - 1) * 16] = p + q;
        }
        __syncthreads();
        for (int r = wv; r < ntop; r += NWAVES) tb[(4 * (r + 1) - 1) * 16] = cb[(4 * (r + 1) - 1) * 16];
        __syncthreads();
    }
    for (int h = top; h >= 1; h >>= 1) {
        const int ne = (N + h) / (2 * h);
        for (int t = wv; t < ne; t += NWAVES) {
            const int e = h * (2 * t + 1) - 1;
            double p = cb[e * 16], q = 0.0;
            if (e - h >= 0) bcr_mv(bcr_frag(F, e, BcrFmt::OLBL, lane), tb[(e - h) * 16], p, q);
            if (e + h < N) bcr_mv(bcr_frag(F, e, BcrFmt::OLBR, lane), tb[(e + h) * 16], p, q);
            tb[e * 16] = p + q;
        }
        __syncthreads();
    }
}

// compile-time helpers of the register-resident schedule (mpcqp_lat.h)
template <int I, int END, class Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
    if constexpr (I < END) { f(std::integral_constant<int, I>{}); static_for<I + 1, END>(f); }
}
constexpr int bcr_ne(int N, int h) { return (N + h) / (2 * h); }          // eliminated stages e = h (2t+1) - 1 < N of the level with stride h
constexpr int bcr_nk(int N, int h) { return N / (2 * h); }                // kept stages i = 2h (t+1) - 1 < N
constexpr int bcr_levels(int N) { int l = 0; for (int h = 1; h - 1 < N; h <<= 1) ++l; return l; }

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

// The dense top of the round on the VECTOR ALU (mpcqp_latw.h says why and what was measured): 0 = on the matrix cores (fragments), 1 = lane (i = lane & 15,
// p = lane >> 4) of the wave that owns block row r holds row 16 r + i times columns [4 nt p, 4 nt (p + 1)), 2 = lane (q = lane >> 3, p = lane & 7) rows 16 r + 2 q,
// 16 r + 2 q + 1 times columns [2 nt p, 2 nt (p + 1)).  Order of the inverse in 16-byte pairs: pair ((r * 2 nt + k) * 64 + lane), k < 2 nt, holds
//   mode 1: T[16 r + (lane & 15)][4 nt (lane >> 4) + 2 k .. + 1]          mode 2: T[16 r + 2 (lane >> 3) + k / nt][2 nt (lane & 7) + 2 (k % nt) .. + 1]
// -- a wave's read k is 64 consecutive pairs (conflict-free).  bcr_topv_rc: (row, first column) of a pair.
#ifndef LATW_TOP_VALU
#define LATW_TOP_VALU 2
#endif
__host__ __device__ inline void bcr_topv_rc(int nt, int pair, int &row, int &col) {
    const int ln = pair & 63, rk = pair >> 6, r = rk / (2 * nt), k = rk - r * 2 * nt;
#if LATW_TOP_VALU == 2
    const int rr = k / nt, kc = k - rr * nt;
    row = 16 * r + 2 * (ln >> 3) + rr; col = 2 * nt * (ln & 7) + 2 * kc;
#else
    row = 16 * r + (ln & 15); col = 4 * nt * (ln >> 4) + 2 * k;
#endif
}

struct BcrFmt {
    static constexpr int NN = 256;                            // doubles per 16 x 16 fragment
    static constexpr int REC = 5 * NN;                        // [ D^-1 | -LbL | -LbR | -LbL' | -LbR' ]
    static constexpr int ODINV = 0, OLBL = NN, OLBR = 2 * NN, OLBLT = 3 * NN, OLBRT = 4 * NN;
    static constexpr int WSTAGE = 4 * NN;                     // global workspace per stage of the factorization: K_ii | K_{i,next} | dK_L | dK_R
    static constexpr int LDSW = 4 * 5 * NN;                   // LDS of the factorization: four groups x [ D | B_L | B_R | Lb_L | Lb_R ]
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
    static constexpr int top_lds(int N) { return 16 * top_count(N) * (16 * top_count(N) + 1) + 2 * 16 * top_count(N); }      // LDS of the inversion: matrix (odd row stride), pivot row, pivot column
};

// ------------------------------------------------------------------------------------------------
// Factorization (run time N).  Wg: this instance's global workspace, N * WSTAGE doubles; W: LDS, BcrFmt::LDSW doubles.
// Four thread groups (one wave each) eliminate four stages of a level side by side; barriers are workgroup-wide, every
// thread makes the same calls.  Returns 1 on a non-positive pivot.
// N = L.bcr >= L.N is the stage count of the elimination tree (the register-resident schedule of mpcqp_lat.h exists for a few sizes): the
// stages beyond the problem's own are identity blocks without couplings -- they cost the schedule a few idle mat-vecs and change nothing.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int factor_bcr(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *Wg, double *W, int *iflag) {
    constexpr int NN = BcrFmt::NN, G = 4, T = NT / G, EPT = NN / T, NB = 16;
    const Lay &L = c.L;
    const int N = L.bcr, NR = L.N, tid = threadIdx.x, g = tid / T, lt = tid % T;
    double *Kd = Wg, *Up = Wg + (size_t)N * NN, *dKL = Up + (size_t)N * NN, *dKR = dKL + (size_t)N * NN;
    if (tid == 0) *iflag = 0;
    // (omega and s through an LDS copy behind the workspace: every entry of a diagonal block sums nx products of them)
    double *oml = W + BcrFmt::LDSW, *svl = oml + L.m;
    for (int r = tid; r < L.m; r += NT) oml[r] = om[r];
    for (int r = tid; r < L.n; r += NT) svl[r] = sv[r];
    __syncthreads();
    for (int idx = tid; idx < N * NN; idx += NT) {
        const int k = idx / NN, r = idx % NN, a = r / NB, b = r % NB;
        Kd[idx] = k < NR ? kkt_diag_entry(c, oml, svl, cc, k, a, b) : (a == b ? 1.0 : 0.0);
        Up[idx] = (k + 1 < NR) ? kkt_sub_entry(c, oml, cc, k, b, a) : 0.0;     // K_{k,k+1}[a][b] = K_{k+1,k}[b][a]
    }
    __syncthreads();
    double *D = W + g * 5 * NN, *BL = D + NN, *BR = D + 2 * NN, *LL = D + 3 * NN, *LR = D + 4 * NN;
    for (int h = 1; h - 1 < N; h <<= 1) {
        if (L.bcrtop && h == 4) break;                        // (dense top: what is left after levels 0 and 1 is inverted explicitly below)
        const int ne = (N + h) / (2 * h);                     // stages e = h (2t+1) - 1 < N
        for (int t0 = 0; t0 < ne; t0 += G) {
            const int t = t0 + g;
            const bool on = t < ne;
            const int e = h * (2 * t + 1) - 1, i1 = e - h, i2 = e + h;
            const bool hasL = on && i1 >= 0, hasR = on && i2 < N;
            if (on) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int idx = lt + T * u, a = idx / NB, b = idx % NB;
                    D[idx] = Kd[(size_t)e * NN + idx];
                    BL[idx] = hasL ? Up[(size_t)i1 * NN + b * NB + a] : 0.0;     // K_{e,i1} = K_{i1,e}'
                    BR[idx] = hasR ? Up[(size_t)e * NN + idx] : 0.0;            // K_{e,i2}
                }
            }
            __syncthreads();
            // in-place Gauss-Jordan inversion of the SPD block (as in factor_all): step p uses the OLD pivot row and column
            for (int pv = 0; pv < NB; ++pv) {
                double rip[EPT], rpj[EPT], d = 1.0;
                if (on) {
                    d = D[pv * NB + pv];
#pragma unroll
                    for (int u = 0; u < EPT; ++u) { const int idx = lt + T * u; rip[u] = D[(idx / NB) * NB + pv]; rpj[u] = D[pv * NB + (idx % NB)]; }
                }
                if (on && !(d > 0.0)) { if (lt == 0) *iflag = 1; d = 1e-300; }
                __syncthreads();
                if (on) {
                    const double inv = 1.0 / d;
#pragma unroll
                    for (int u = 0; u < EPT; ++u) {
                        const int idx = lt + T * u, i = idx / NB, j = idx % NB;
                        D[idx] = (i == pv) ? (j == pv ? inv : rpj[u] * inv) : (j == pv ? -rip[u] * inv : D[idx] - rip[u] * rpj[u] * inv);
                    }
                }
                __syncthreads();
            }
            double sy[EPT];
            if (on) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) { const int idx = lt + T * u, a = idx / NB, b = idx % NB; sy[u] = 0.5 * (D[a * NB + b] + D[b * NB + a]); }
            }
            __syncthreads();
            if (on) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) D[lt + T * u] = sy[u];
            }
            __syncthreads();
            if (on) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) {              // Lb = D^-1 B
                    const int idx = lt + T * u, a = idx / NB, b = idx % NB;
                    double accL = 0.0, accR = 0.0;
#pragma unroll 8
                    for (int l = 0; l < NB; ++l) { const double dv = D[a * NB + l]; accL += dv * BL[l * NB + b]; accR += dv * BR[l * NB + b]; }
                    LL[idx] = accL; LR[idx] = accR;
                }
            }
            __syncthreads();
            if (on) {
                double *rec = F + (size_t)e * BcrFmt::REC;
                const int nbk = e >= NR ? 0 : (e < L.NcT) ? L.nb : L.nx;     // variables of this stage; the rest is padding (identity in K, zero in the factor)
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int idx = lt + T * u, a = idx / NB, b = idx % NB;
                    const int fp = frag_pos<NB>(a, b);
                    rec[BcrFmt::ODINV + fp] = (a >= nbk || b >= nbk) ? 0.0 : D[idx];
                    rec[BcrFmt::OLBL + fp] = -LL[idx];
                    rec[BcrFmt::OLBR + fp] = -LR[idx];
                    rec[BcrFmt::OLBLT + fp] = -LL[b * NB + a];
                    rec[BcrFmt::OLBRT + fp] = -LR[b * NB + a];
                    double sL = 0.0, sR = 0.0, sU = 0.0;      // K_{i1,e} Lb_L,  K_{i2,e} Lb_R,  K_{i1,e} Lb_R
#pragma unroll 8
                    for (int l = 0; l < NB; ++l) {
                        const double bl = BL[l * NB + a], br = BR[l * NB + a];
                        sL += bl * LL[l * NB + b]; sR += br * LR[l * NB + b]; sU += bl * LR[l * NB + b];
                    }
                    dKL[(size_t)e * NN + idx] = sL;
                    dKR[(size_t)e * NN + idx] = sR;
                    if (hasL && hasR) Up[(size_t)i1 * NN + idx] = -sU;          // the coupling i1 <-> i2 of the next level
                }
            }
            __syncthreads();
        }
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
        // ---- the top: stages i_r = 4 (r + 1) - 1, r < nt, with the Schur complements the two levels left in Kd (diagonal blocks) and Up (Up[i] =
        // K_{i,i+4} for a kept i).  Assembled dense in LDS (odd row stride), inverted in place by Gauss-Jordan sweeps -- SPD: no pivoting, a
        // non-positive pivot is reported as everywhere else -- with thread t owning column t mod 128 of every (NT / 128)-th row, then written as
        // fragments with zero rows and columns where a stage has no variable (as D^-1 above).
        const int nt = L.bcrtop, NRt = nt * NB, ld = NRt + 1;
        double *M = W, *prow = M + NRt * ld, *pcol = prow + NRt;
        for (int idx = tid; idx < NRt * NRt; idx += NT) {
            const int r = idx / NRt, cx = idx - r * NRt, br = r / NB, a = r % NB, bc = cx / NB, b = cx % NB;
            const int i = 4 * (br + 1) - 1, j = 4 * (bc + 1) - 1;
            double v = 0.0;
            if (br == bc) v = Kd[(size_t)i * NN + a * NB + b];
            else if (bc == br + 1) v = Up[(size_t)i * NN + a * NB + b];          // K_{i,j}, j = i + 4
            else if (bc + 1 == br) v = Up[(size_t)j * NN + b * NB + a];          // K_{i,j} = K_{j,i}'
            M[r * ld + cx] = v;
        }
        __syncthreads();
        const int jc = tid & 127, i0 = tid >> 7;
        constexpr int RG = NT / 128;                          // row groups
        for (int pv = 0; pv < NRt; ++pv) {
            for (int t = tid; t < NRt; t += NT) { prow[t] = M[pv * ld + t]; pcol[t] = M[t * ld + pv]; }
            __syncthreads();
            double d = prow[pv];
            if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
            const double inv = 1.0 / d;
            if (jc < NRt) {
                const double rpj = prow[jc];
                for (int i = i0; i < NRt; i += RG) {
                    const double rip = pcol[i], cur = M[i * ld + jc];
                    M[i * ld + jc] = (i == pv) ? (jc == pv ? inv : rpj * inv) : (jc == pv ? -rip * inv : cur - rip * rpj * inv);
                }
            }
            __syncthreads();
        }
        double *Ft = F + BcrFmt::top_off(N);
        auto topval = [&](int br, int a, int bc, int b) {      // entry (16 br + a, 16 bc + b) of the symmetrised inverse, zero where a stage has no variable
            const int i = 4 * (br + 1) - 1, j = 4 * (bc + 1) - 1;
            const int nbi = i >= NR ? 0 : (i < L.NcT) ? L.nb : L.nx, nbj = j >= NR ? 0 : (j < L.NcT) ? L.nb : L.nx;
            return (a >= nbi || b >= nbj) ? 0.0 : 0.5 * (M[(br * NB + a) * ld + bc * NB + b] + M[(bc * NB + b) * ld + br * NB + a]);
        };
        for (int idx = tid; idx < nt * nt * NN; idx += NT) {
            const int blk = idx / NN, e = idx - blk * NN, a = e / NB, b = e % NB, br = blk / nt, bc = blk - br * nt;
            Ft[(size_t)blk * NN + frag_pos<NB>(a, b)] = topval(br, a, bc, b);
        }
#if LATW_TOP_VALU
        double *Fv = F + BcrFmt::topv_off(N);                 // the same in the round's row-part order
        for (int idx = tid; idx < nt * nt * (NN / 2); idx += NT) {
            int row, col; bcr_topv_rc(nt, idx, row, col);
            Fv[2 * (size_t)idx] = topval(row / NB, row % NB, col / NB, col % NB);
            Fv[2 * (size_t)idx + 1] = topval(row / NB, row % NB, (col + 1) / NB, (col + 1) % NB);
        }
#endif
        if (tid == 0) iflag[2] = 0;                          // (the rounds' LDS copy of the top is stale -- and this workspace has just run over it: admm_latw)
        __syncthreads();
    }
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

// mpcqp_dense.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Small problems (the reference's own examples: point mass, cart-pole -- N (nx+nu) <= 128): the KKT solve of an ADMM iteration
// as ONE dense mat-vec with the explicit inverse of the reduced KKT matrix, held in REGISTERS for a whole round.
//
// Why: the block-tridiagonal sweeps are a chain of N dependent stage steps (~250 shader cycles each, twice per iteration);
// a single small QP therefore spends 8 us per ADMM iteration on one CU while 255 CUs idle -- three times slower than one CPU
// core.  For NR = N (nx + nu) <= 128 unknowns the inverse K^-1 is NR^2 <= 16384 doubles = 64 per thread: every thread keeps one
// half-row of K^-1 in 128 VGPRs, the right-hand side is read from LDS (wave-uniform addresses: broadcasts), and the "solve" is
// 64 FMAs per thread plus one exchange of the two half-row sums -- a few hundred cycles, no dependent chain, no factor stream.
// Per round the instance reads its K^-1 once (128 KB); per iteration nothing.
//
// Variable order of the dense system: stage-major COMPACT, v = k (nx + nu) + a (the padded order of Tc without the padding).
// Factor record (DenseFmt): F[j * NT + t] = K^-1[row(t)][64 half(t) + j],  row(t) = t >> 1, half(t) = t & 1, j < 64 --
// the order the threads load it in (coalesced), zero beyond NR and in rows/columns of absent inputs (stages >= NcT carry no u).
// The two halves of a row sit in NEIGHBOURING LANES: their sums meet through one DPP swap, not through LDS and a barrier.
#pragma once

struct DenseFmt {
    static constexpr int ROWS = 128, JW = 64;                 // row slots, columns per thread
    static constexpr int DOUBLES = JW * NT;                   // doubles per instance
    static constexpr int HPAD = 2;                            // the second half of the compact vector starts HPAD doubles late: lanes
                                                              // of the two halves read cv[j] and cv[64 + HPAD + j] in one instruction -- other banks
    static constexpr int CV = ROWS + HPAD + 6;                // compact right-hand side (16-byte multiples)
    static constexpr int SCRATCH = 4 * ROWS + 64;             // LDS doubles: rhs | solution | two exchange vectors of the small-problem iteration
};
__device__ __forceinline__ int dense_cv_index(int v) { return v + (v >= DenseFmt::JW ? DenseFmt::HPAD : 0); }
// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double lane_swap1(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)xi, 0xB1, 0xF, 0xF, false), hi = __builtin_amdgcn_mov_dpp((int)(xi >> 32), 0xB1, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
static_assert(kLatOnly || NT == 256, "the dense path maps 128 rows x 2 column halves onto 256 threads");

__device__ __forceinline__ bool dense_dead(const Lay &L, int v) {       // variable slot v = (k, a) without a variable
    const int k = v / L.nb, a = v - k * L.nb;
    return a >= L.nx && k >= L.Nc;                                      // (Nc < Np: the held input u_{Nc-1} is an ordinary unknown of the dense system)
}

__device__ __forceinline__ double kkt_entry_generic(const Ctx &c, const double *om, const double *sv, double cc, int v, int w);      // mpcqp_border.h

// K (dense, SPD) assembled from the stage blocks (only the block-tridiagonal band is evaluated; omega and s are read from an LDS copy: an
// entry sums nx products of them), inverted in place by Gauss-Jordan sweeps (NR steps).  A thread keeps its half row -- up to 64 entries --
// in registers through the steps: a step reads the pivot row (64 broadcast reads, issued together) and its one pivot-column entry from the
// copies made before the barrier, updates the registers and writes the half row back for the next step's copies.
// W: LDS, NR * ld + 2 * ROWS + m + n doubles (ld = L.dld, odd: column accesses are bank-conflict free).  Returns 1 on a non-positive pivot.
__device__ __forceinline__ int factor_dense(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag) {
    const Lay &L = c.L;
    const int tid = threadIdx.x, NR = L.NR, ld = L.dld, nb = L.nb;
    double *prow = W + NR * ld, *pcol = prow + DenseFmt::ROWS, *oml = pcol + DenseFmt::ROWS, *svl = oml + L.m;
    if (tid == 0) *iflag = 0;
    TICK_RESET
    TICK_START
    for (int r = tid; r < L.m; r += NT) oml[r] = om[r];
    for (int r = tid; r < L.n; r += NT) svl[r] = sv[r];
    for (int e = tid; e < NR * ld; e += NT) W[e] = 0.0;
    __syncthreads();
    const int per = 3 * nb * nb;                              // entries of one block row of the band: [sub | diag | super]
    for (int e = tid; e < L.N * per; e += NT) {
        const int ki = e / per, r = e - ki * per, blk = r / (nb * nb), q = r - blk * nb * nb, ai = q / nb, aj = q - ai * nb;
        const int kj = ki + blk - 1;
        if (kj < 0 || kj >= L.N) continue;
        double v;
        if (blk == 1) v = kkt_diag_entry(c, oml, svl, cc, ki, ai, aj);
        else if (blk == 0) v = kkt_sub_entry(c, oml, cc, kj, ai, aj);          // K_{ki,ki-1}
        else v = kkt_sub_entry(c, oml, cc, ki, aj, ai);                        // K_{ki,ki+1} = K_{ki+1,ki}'
        W[(ki * nb + ai) * ld + kj * nb + aj] = v;
    }
    __syncthreads();
    if (L.border) {
        // Nc < Np: the held input u_{Nc-1} couples to every later stage (mpc.py:513-517,540-543).  The sweeps treat it by bordering; a dense K
        // simply has those entries: its rows and columns, taken entry by entry from the matrix-free operators (the band pass above saw stage
        // Nc-1 without an input).
        const int vb0 = (L.Nc - 1) * nb + L.nx, ub0 = L.ou + (L.Nc - 1) * L.nu;
        for (int e = tid; e < L.nu * NR; e += NT) {
            const int jj = e / NR, w = e - jj * NR, kw = w / nb, aw = w - kw * nb;
            const int var = aw < L.nx ? kw * L.nx + aw : (kw < L.Nc ? L.ou + kw * L.nu + (aw - L.nx) : -1);
            const double v = var >= 0 ? kkt_entry_generic(c, oml, svl, cc, ub0 + jj, var) : 0.0;
            W[(vb0 + jj) * ld + w] = v;
            // (the mirror entry -- except inside the held input's own nu x nu block, where the item (jj', w') of the other input writes it: two
            //  threads storing sums formed in different orders to one address made the factor, and with it every iterate, depend on which store
            //  landed last -- 1e-13 from handle to handle at nu = 2, found by scripts/fuzz_loop.py)
            if (w < vb0 || w >= vb0 + L.nu) W[w * ld + vb0 + jj] = v;
        }
        __syncthreads();
    }
    TICK(0)
    const int i = tid & (DenseFmt::ROWS - 1), h = tid >> 7;
    const int j0 = DenseFmt::JW * h;
    const bool live = i < NR;
    const int nvalid = __builtin_amdgcn_readfirstlane(min(DenseFmt::JW, NR - j0));      // columns of this half that exist (the same for a whole wave)
    double cur[DenseFmt::JW];
#pragma unroll
    for (int u = 0; u < DenseFmt::JW; ++u) cur[u] = (live && j0 + u < NR) ? W[i * ld + j0 + u] : 0.0;
    for (int pv = 0; pv < NR; ++pv) {
        if (tid < NR) { prow[tid] = W[pv * ld + tid]; pcol[tid] = W[tid * ld + pv]; }
        __syncthreads();
        double d = prow[pv];
        if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
        const double inv = 1.0 / d;
        if (live) {
            // (in chunks of 16 columns: the pivot-row values of a chunk are read together; chunks beyond NR are skipped; the two threads of the
            //  pivot row take their own branch instead of a select per entry.  nvalid is wave-uniform: scalar branches, no exec masking)
            const double f = pcol[i] * inv;
#pragma unroll
            for (int ch = 0; ch < DenseFmt::JW / 16; ++ch) {
                if (16 * ch < nvalid) {
                double pr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pr[u] = prow[j0 + 16 * ch + u];      // (entries beyond NR are never stored)
                if (i == pv) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) cur[16 * ch + u] = (j0 + 16 * ch + u == pv) ? inv : pr[u] * inv;
                } else {
#pragma unroll
                    for (int u = 0; u < 16; ++u) cur[16 * ch + u] = (j0 + 16 * ch + u == pv) ? -f : fma(-f, pr[u], cur[16 * ch + u]);
                }
                double *wr = W + i * ld + j0 + 16 * ch;
                if (16 * ch + 16 <= nvalid) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) wr[u] = cur[16 * ch + u];
                } else {
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (16 * ch + u < nvalid) wr[u] = cur[16 * ch + u];
                }
                }
            }
        }
        __syncthreads();
    }
    TICK(1)
    const int ro = tid >> 1, co = DenseFmt::JW * (tid & 1);   // register order: row, first column of this thread
    const bool dead_r = ro >= NR || dense_dead(L, ro);
    for (int j = 0; j < DenseFmt::JW; ++j) {
        const int cidx = co + j;
        double v = 0.0;
        if (!dead_r && cidx < NR) {
            // (Nc < Np: the column of an input slot behind the held input is a copy of the held input's column -- admm_tiny puts the slot's
            //  share of the held input's A'W into the right-hand side there, the mat-vec adds the shares up)
            const int kc = cidx / nb, ac = cidx - kc * nb;
            const int src = (L.border && ac >= L.nx && kc > L.Nc) ? (L.Nc - 1) * nb + ac : cidx;
            if (!dense_dead(L, src)) v = 0.5 * (W[ro * ld + src] + W[src * ld + ro]);
        }
        F[(size_t)j * NT + tid] = v;
    }
    TICK(2)
    TICK_FLUSH
    __syncthreads();
    return *iflag;
}

// This thread's half-row of K^-1 (the round's resident copy).
__device__ __forceinline__ void dense_load(const double *F, double *Kreg) {
    cgdouble *Fg = (cgdouble *)F;
#pragma unroll
    for (int j = 0; j < DenseFmt::JW; ++j) Kreg[j] = Fg[(size_t)j * NT + threadIdx.x];
}

// The half-row dot product of this thread with the compact vector cvec (LDS), both halves summed: every lane pair (2v, 2v+1)
// returns row v of K^-1 cvec.  REGS: K^-1 from the registers loaded by dense_load; else streamed from F.
template <bool REGS>
__device__ __forceinline__ double dense_row(const double *Kreg, const double *F, const double *cvec) {
    constexpr int BATCH = 16;                                 // LDS values in flight
    const int tid = threadIdx.x;
    const double *cv = cvec + (tid & 1) * (DenseFmt::JW + DenseFmt::HPAD);
    cgdouble *Fg = (cgdouble *)F;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // The reads are issued in groups and fenced: left alone, a scheduler that is short of registers (the inverse takes 128 of
    // them) waits for every single read before the two FMAs that use it -- 32 LDS round trips per mat-vec instead of 4.
    // 8-byte reads on purpose: every lane of a half reads the SAME address, which the LDS serves as a broadcast for 4- and 8-byte
    // reads; 16-byte reads of one address by 32 lanes are serialised (measured: 260 cycles per instruction).
#pragma unroll
    for (int jb = 0; jb < DenseFmt::JW; jb += BATCH) {
        double c[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) c[q] = cv[jb + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BATCH; q += 4) {
            const int j = jb + q;
            const double k0 = REGS ? Kreg[j] : Fg[(size_t)j * NT + tid], k1 = REGS ? Kreg[j + 1] : Fg[(size_t)(j + 1) * NT + tid];
            const double k2 = REGS ? Kreg[j + 2] : Fg[(size_t)(j + 2) * NT + tid], k3 = REGS ? Kreg[j + 3] : Fg[(size_t)(j + 3) * NT + tid];
            a0 = fma(k0, c[q], a0); a1 = fma(k1, c[q + 1], a1); a2 = fma(k2, c[q + 2], a2); a3 = fma(k3, c[q + 3], a3);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const double part = (a0 + a1) + (a2 + a3);
    return part + lane_swap1(part);
}

// Tc <- K^-1 Tc on the padded stage-major vector Tc (stride NB): gather to compact order, row dot products, scatter
// (verification kernel).  myidx: Tc index of compact slot threadIdx.x (threads < NR).  scr: LDS, DenseFmt::CV doubles.
template <int NB>
__device__ __forceinline__ void dense_core(const Lay &L, const double *F, double *Tc, double *scr, int myidx) {
    const int tid = threadIdx.x;
    if (tid < DenseFmt::ROWS) scr[dense_cv_index(tid)] = tid < L.NR ? Tc[myidx] : 0.0;
    __syncthreads();
    const double r = dense_row<false>(nullptr, F, scr);
    const int v = tid >> 1, k = v / L.nb;
    if (!(tid & 1) && v < L.NR) Tc[k * NB + (v - k * L.nb)] = r;
    __syncthreads();
}
template <int NB>
__device__ __forceinline__ int dense_slot(const Lay &L) {    // Tc index of this thread's compact slot
    const int t = threadIdx.x < L.NR ? (int)threadIdx.x : 0;
    const int k = t / L.nb;
    return k * NB + (t - k * L.nb);
}

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
// Factor record (DenseFmt): F[j * NT + t] = K^-1[2 (t >> 2) + (j >> 5)][32 (t & 3) + (j & 31)], j < 64 -- a QUAD of lanes holds two rows, each lane a
// quarter of their columns, in the order the threads load it (coalesced); zero beyond NR and in rows/columns of absent inputs (stages >= NcT carry
// no u).  (Until round 6 a lane pair held one row, each lane half of it: 64 broadcast reads of the right-hand side per lane and mat-vec -- 128 KB
// through the LDS return path per iteration, 1 024 of the mat-vec's 1 314 cycles.  Two rows per lane use every value read twice: 32 reads.)
// The four partial sums of a row meet through two DPP steps inside the quad, not through LDS and a barrier; the lane pair (2v, 2v+1) ends up with row v.
#pragma once

struct DenseFmt {
    static constexpr int ROWS = 128, JW = 64;                 // row slots, columns per thread
    static constexpr int DOUBLES = JW * NT;                   // doubles per instance
    static constexpr int QW = 32, HPAD = 2;                   // columns per lane and row; quarter p of the compact vector starts p * HPAD doubles late: the
                                                              // lanes of the four quarters read cv[j + (QW + HPAD) p] in one instruction -- four different banks
    static constexpr int CV = ROWS + 3 * HPAD + 2;            // compact right-hand side (16-byte multiples; the last entry is never written)
    static constexpr int SCRATCH = 4 * ROWS + 64;             // LDS doubles: rhs | solution | two exchange vectors of the small-problem iteration
};
__device__ __forceinline__ int dense_cv_index(int v) { return v + DenseFmt::HPAD * (v / DenseFmt::QW); }
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
    for (int j = 0; j < DenseFmt::JW; ++j) {                  // register order (DenseFmt): rows 2 (tid >> 2), + 1; columns 32 (tid & 3) ..
        const int ro = 2 * (tid >> 2) + (j >> 5), cidx = DenseFmt::QW * (tid & 3) + (j & 31);
        double v = 0.0;
        if (ro < NR && !dense_dead(L, ro) && cidx < NR) {
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

template <int CTRL>
__device__ __forceinline__ double quad_perm(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)xi, CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_mov_dpp((int)(xi >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// This thread's two quarter-rows times its quarter of the compact vector cvec (LDS), summed over the quad: every lane pair (2v, 2v+1)
// returns row v of K^-1 cvec.  REGS: K^-1 from the registers loaded by dense_load; else streamed from F.
template <bool REGS>
__device__ __forceinline__ double dense_row(const double *Kreg, const double *F, const double *cvec) {
    constexpr int BATCH = 16, QW = DenseFmt::QW;              // LDS values in flight
    const int tid = threadIdx.x;
    const double *cv = cvec + (tid & 3) * (QW + DenseFmt::HPAD);
    cgdouble *Fg = (cgdouble *)F;
    auto kk = [&](int j) { return REGS ? Kreg[j] : Fg[(size_t)j * NT + tid]; };
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    // The reads are issued in groups and fenced: left alone, a scheduler that is short of registers (the inverse takes 128 of
    // them) waits for every single read before the FMAs that use it -- a round trip per read instead of one per group.
    // 8-byte reads on purpose: every lane of a quarter reads the SAME address, which the LDS serves as a broadcast for 4- and 8-byte
    // reads; 16-byte reads of one address by many lanes are serialised (measured: 260 cycles per instruction).
#pragma unroll
    for (int jb = 0; jb < QW; jb += BATCH) {
        double c[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) c[q] = cv[jb + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BATCH; q += 2) {
            const int j = jb + q;
            a0 = fma(kk(j), c[q], a0); a1 = fma(kk(j + 1), c[q + 1], a1);
            b0 = fma(kk(QW + j), c[q], b0); b1 = fma(kk(QW + j + 1), c[q + 1], b1);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double r0 = a0 + a1, r1 = b0 + b1;
    r0 += quad_perm<0xB1>(r0); r1 += quad_perm<0xB1>(r1);    // lane ^ 1
    r0 += quad_perm<0x4E>(r0); r1 += quad_perm<0x4E>(r1);    // lane ^ 2: every lane of the quad holds both rows' sums (the same bits)
    return (tid & 2) ? r1 : r0;
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

// mpcqp_group.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// GROUPED stages: the KKT backend for long horizons of SMALL stages (nx + nu <= 8) -- the reference's own published timing example is one
// (examples/example_inverted_pendulum_kalman.ipynb: nx = 4, nu = 1, Np = 150, Nc = 75).
//
// The twisted block LDL' (mpcqp_factor.h / mpcqp_sweeps.h) spends one 16 x 16 block and one dependent step of its sweeps per STAGE, however few
// of the sixteen slots the stage fills: at nx + nu = 5 the chain is 151 steps of blocks that are 90 % padding -- 17 us of a 25 us iteration for one
// controller, 2.3 x slower than one CPU core (round 3).  Here g = floor(16 / (nx + nu)) consecutive stages share a block ("super-stage": slot
// A = i (nx + nu) + a is element a of sub-stage i).  The reduced KKT matrix is block tridiagonal over super-stages as well -- the last sub-stage of
// one couples to the first sub-stage of the next, everything else is inside a block -- so the same twisted elimination applies with N' = ceil(N / g)
// steps: a third of the chain and a third of the factor bytes at g = 3.
//
// The blocks are no longer "weights times a constant [Ad Bd]" (a diagonal block holds g - 1 stage couplings itself), so nothing is applied
// matrix-free: per super-stage the factor keeps three full fragments
//       [ -Mh_K | -Mh_K' | S_K^-1 ]      Mh_K = K_{K,nbr} S_nbr^-1 (nbr = K-1 in the top half, K+1 in the bottom half),  S_K = K_KK - Mh_K K_{K,nbr}'
// and the solve is   forward  yh_K = b_K - Mh_K yh_nbr,  w_K = S_K^-1 yh_K   (one dependent mat-vec per step; w_K rides in its shadow),
//                    middle   x_m = S_m^-1 (b_m - Mh_m yh_{m-1} - Mt_m yh_{m+1})          (Mt_m: record N')
//                    backward x_K = w_K - Mh_next' x_next          (next = the stage towards the middle: one dependent mat-vec per step).
// Everything around the solve -- the parallel phases, the bordered correction of a held input (Nc < Np), the residual evaluation -- keeps the
// stage-major layout Tc[k * 16 + a]: kkt_core repacks into the grouped vector Tg[K * 16 + A] (behind Tc in the work area) and back, two LDS
// passes over N (nx + nu) values.  Dead slots (block padding, sub-stages beyond the horizon, inputs beyond the control horizon) are identity in K
// and zero in the stored inverse, like everywhere else.
#pragma once

// prefetch ring depths of the two sweeps (stages in flight in registers: 16 VGPRs each forward, 8 backward).  Measured on the notebook shape, one
// controller: 0.19 us per forward step, 0.11 per backward step; deeper rings (4 / 8, 5 / 8) do NOT help -- update() 461 us against 452 -- so the steps
// are not waiting for these loads.
#ifndef MPCQP_GRP_FWD_DEPTH
#define MPCQP_GRP_FWD_DEPTH 3
#endif
#ifndef MPCQP_GRP_BWD_DEPTH
#define MPCQP_GRP_BWD_DEPTH 4
#endif
struct GroupFmt {
    static constexpr int NN = 256;
    static constexpr int OMH = 0, OMHT = NN, OSINV = 2 * NN, REC = 3 * NN;      // (the forward matrix first: chain_sweep reads a stage's record from its start)
};
__host__ __device__ inline int group_size(int nb) { return 16 / nb; }
__host__ __device__ inline int group_count(int N, int g) { return (N + g - 1) / g; }

// element (A, B) of the diagonal block of super-stage K
__device__ __forceinline__ double grp_diag_entry(const Ctx &c, const double *om, const double *sv, double cc, int g, int K, int A, int B) {
    const Lay &L = c.L;
    const int nb = L.nb;
    if (A >= g * nb || B >= g * nb) return A == B ? 1.0 : 0.0;
    const int i = A / nb, a = A - i * nb, j = B / nb, b = B - j * nb, ki = K * g + i, kj = K * g + j;
    if (ki >= L.N || kj >= L.N) return A == B ? 1.0 : 0.0;
    if (i == j) return kkt_diag_entry<true>(c, om, sv, cc, ki, a, b);
    if (i == j + 1) return kkt_sub_entry(c, om, cc, kj, a, b);             // K_{ki,ki-1}[a][b]
    if (j == i + 1) return kkt_sub_entry(c, om, cc, ki, b, a);             // its mirror
    return 0.0;
}
// element (A, B) of the coupling block K_{K,K-1}: rows in super-stage K (only its first sub-stage has any), columns in K-1 (only its last)
__device__ __forceinline__ double grp_sub_entry(const Ctx &c, const double *om, double cc, int g, int K, int A, int B) {
    const Lay &L = c.L;
    const int nb = L.nb;
    if (A >= nb || B < (g - 1) * nb || B >= g * nb) return 0.0;
    const int k = K * g;
    if (k < 1 || k >= L.N) return 0.0;
    return kkt_sub_entry(c, om, cc, k - 1, A, B - (g - 1) * nb);
}
__device__ __forceinline__ bool grp_dead(const Lay &L, int g, int K, int A) {
    if (A >= g * L.nb) return true;
    const int i = A / L.nb, a = A - i * L.nb, k = K * g + i;
    return k >= L.N || (a >= L.nx && k >= L.NcT);
}

// Factorization: one super-stage at a time by the whole workgroup (one entry per thread), twisted order.  W: LDS, 5 * 256 doubles.
__device__ __forceinline__ int factor_group(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag) {
    constexpr int NN = GroupFmt::NN, NB = 16;
    static_assert(NT >= NN, "one entry of a 16 x 16 block per thread (threads beyond 256 -- the 512-thread kernels of mpcqp_w8.hip -- only keep the barriers company)");
    const Lay &L = c.L;
    const int g = L.grp, NS = group_count(L.N, g), mid = NS / 2, tid = threadIdx.x;
    const bool on = tid < NN;                                   // this thread owns an entry
    const int et = on ? tid : 0, A = et / NB, B = et % NB;
    double *S = W, *C = W + NN, *Mh = W + 2 * NN, *SnA = W + 3 * NN, *SnB = W + 4 * NN;
    if (tid == 0) *iflag = 0;
    // omega and s through an LDS copy behind the work matrices where the work area holds them (every entry of a block reads several of them,
    // one global round trip each: they were a good part of a stage's 9 us)
    if (5 * NN + L.m + L.n <= L.tsz) {
        double *oml = W + 5 * NN, *svl = oml + L.m;
        for (int r = tid; r < L.m; r += NT) oml[r] = om[r];
        for (int r = tid; r < L.n; r += NT) svl[r] = sv[r];
        om = oml; sv = svl;
    }
    __syncthreads();
    // one stage: S_K and its inverse; rec: where -Mh / -Mh' of the neighbour above (up) or below go
    auto stage = [&](int K, const double *SnU, const double *SnD, double *SnOut, double *recU, double *recD) {
        if (on) S[tid] = grp_diag_entry(c, om, sv, cc, g, K, A, B);
        for (int side = 0; side < 2; ++side) {
            const double *Sn = side == 0 ? SnU : SnD;
            double *rec = side == 0 ? recU : recD;
            if (!Sn) continue;                                      // (uniform)
            __syncthreads();
            if (on) C[tid] = side == 0 ? grp_sub_entry(c, om, cc, g, K, A, B) : grp_sub_entry(c, om, cc, g, K + 1, B, A);      // K_{K,K-1} / K_{K,K+1} = K_{K+1,K}'
            __syncthreads();
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < NB; ++l) acc = fma(C[A * NB + l], Sn[l * NB + B], acc);
            if (on) {
                Mh[tid] = acc;
                rec[GroupFmt::OMH + frag_pos<NB>(A, B)] = -acc;
                rec[GroupFmt::OMHT + frag_pos<NB>(B, A)] = -acc;
            }
            __syncthreads();
            double sub = 0.0;
#pragma unroll
            for (int l = 0; l < NB; ++l) sub = fma(Mh[A * NB + l], C[B * NB + l], sub);
            if (on) S[tid] -= sub;
        }
        __syncthreads();
        // in-place Gauss-Jordan inversion (SPD block; step p reads the matrix step p-1 wrote and writes the other buffer: one barrier per step)
        double cur = S[et];
        for (int pv = 0; pv < NB; ++pv) {
            const double *Sr = (pv & 1) ? C : S;
            double *Sw = (pv & 1) ? S : C;
            double d = Sr[pv * NB + pv];
            const double rip = Sr[A * NB + pv], rpj = Sr[pv * NB + B];
            if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
            const double inv = 1.0 / d, t = rip * inv;
            const bool rowp = A == pv, colp = B == pv;
            const double off = rowp ? rpj * inv : fma(-t, rpj, cur), onp = rowp ? inv : -t;
            cur = colp ? onp : off;
            if (on) Sw[tid] = cur;
            __syncthreads();
        }
        const double sym = 0.5 * (S[A * NB + B] + S[B * NB + A]);
        __syncthreads();
        if (on) {
            SnOut[tid] = sym;
            F[(size_t)K * GroupFmt::REC + GroupFmt::OSINV + frag_pos<NB>(A, B)] = (grp_dead(L, g, K, A) || grp_dead(L, g, K, B)) ? 0.0 : sym;
        }
        __syncthreads();
    };
    auto rec = [&](int K) { return F + (size_t)K * GroupFmt::REC; };
    // (records that never receive a forward matrix -- the two ends -- hold zeros there)
    for (int e = tid; e < 2 * NN; e += NT) { rec(0)[e] = 0.0; rec(NS - 1)[e] = 0.0; rec(NS)[e] = 0.0; rec(NS)[2 * NN + (e & (NN - 1))] = 0.0; }
    for (int K = 0; K < mid; ++K) stage(K, K > 0 ? SnA : nullptr, nullptr, SnA, rec(K), nullptr);
    for (int K = NS - 1; K > mid; --K) stage(K, nullptr, K < NS - 1 ? SnB : nullptr, SnB, nullptr, rec(K));
    stage(mid, mid > 0 ? SnA : nullptr, mid < NS - 1 ? SnB : nullptr, SnA, rec(mid), rec(NS));      // the middle's second forward matrix: record N'
    return *iflag;
}

// One half of the forward elimination by ONE wave, stages K = first + dir * i, i = 0..nsteps:
//     yh_K = b_K + Fwd(K) yh_{K-dir}   (yh_first = b_first)         the dependent chain: one mat-vec per step
//     w_K  = S_K^-1 yh_K  -> Tg[K]                                    what the back substitution starts from
// and the last yh, which the middle stage needs, goes to Tg[ylast].  A wave issues in order: written right behind yh_K, the product for w_K would
// make the chain wait for it.  It is therefore formed ONE STEP LATER, next to the chain's product for yh_{K+1} -- two independent MFMA
// sequences in one block, which the scheduler interleaves: w rides in the latency of the chain.  Fragments of the next stages are prefetched
// into a register ring (branch-free refills with clamped indices, like chain_sweep).
__device__ __forceinline__ void group_fwd_sweep(const int first, const int dir, const int nsteps, const int ylast, const double *F, double *Tg) {
    constexpr int NB = 16, DEPTH = MPCQP_GRP_FWD_DEPTH;
    const int lane = opaque_lane(threadIdx.x & 63);
    double *tb = Tg + vec_lane_offset(lane);
    const bool writer = MPCQP_STORE_ALL ? true : vec_lane_writer(lane);
    auto stage_of = [&](int i) { return first + dir * (i < nsteps ? i : nsteps); };
    d4 rf[DEPTH], rs[DEPTH];                              // ring slot d: forward matrix of stage i, S^-1 of stage i - 1
    auto ring_load = [&](int i, int d) {
        rf[d] = *(cgd4 *)(F + (size_t)stage_of(i) * GroupFmt::REC + GroupFmt::OMH + lane * 4);
        rs[d] = *(cgd4 *)(F + (size_t)stage_of(i - 1) * GroupFmt::REC + GroupFmt::OSINV + lane * 4);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring_load(1 + d, d);
    double run[1], own[1], nxt[1];
    vec_load<NB>(tb, first, run);
    vec_load<NB>(tb, stage_of(1), own);
    auto step = [&](int i, int d) {
        const int K = stage_of(i);
        vec_load<NB>(tb, stage_of(i + 1), nxt);
        double y[1] = {own[0]}, w[1] = {0.0};
        frag_matvec<NB>(&rf[d], run, y);                          // the chain: yh_K from yh_{K-dir}
        frag_matvec<NB>(&rs[d], run, w);                          // beside it: w_{K-dir} = S^-1 yh_{K-dir}
        vec_store<NB>(tb, K - dir, w, writer);
        run[0] = y[0];
        own[0] = nxt[0];
    };
    int i0 = 1;
    for (; i0 + DEPTH - 1 <= nsteps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { step(i0 + d, d); ring_load(i0 + d + DEPTH, d); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (i0 + d <= nsteps) step(i0 + d, d);
    {   // the last stage of the half: its yh for the middle stage, its w
        const int K = stage_of(nsteps);
        const d4 sl = *(cgd4 *)(F + (size_t)K * GroupFmt::REC + GroupFmt::OSINV + lane * 4);
        double w[1] = {0.0};
        frag_matvec<NB>(&sl, run, w);
        vec_store<NB>(tb, ylast, run, writer);
        vec_store<NB>(tb, K, w, writer);
    }
}

// One half of the back substitution by ONE wave: for i = 1..nsteps, K = first + dir * i:   Tg[K] (= w_K) <- w_K + MhT(K - dir) Tg[K - dir]
// (MhT(J): the transposed forward matrix stored with stage J; `extra` replaces J = first, the middle stage, where the bottom half needs -Mt_m').
__device__ __forceinline__ void group_back_sweep(const int first, const int dir, const int nsteps, const int extra, const double *F, double *Tg) {
    constexpr int NB = 16, DEPTH = MPCQP_GRP_BWD_DEPTH;
    const int lane = opaque_lane(threadIdx.x & 63);
    double *tb = Tg + vec_lane_offset(lane);
    const bool writer = MPCQP_STORE_ALL ? true : vec_lane_writer(lane);
    if (nsteps < 1) return;
    auto stage_of = [&](int i) { return first + dir * (i < nsteps ? i : nsteps); };
    auto src_of = [&](int i) { const int J = stage_of(i) - dir; return (J == first && extra >= 0) ? extra : J; };
    d4 rm[DEPTH];
    auto ring_load = [&](int i, int d) { rm[d] = *(cgd4 *)(F + (size_t)src_of(i) * GroupFmt::REC + GroupFmt::OMHT + lane * 4); };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring_load(1 + d, d);
    double run[1], own[1], nxt[1];
    vec_load<NB>(tb, first, run);
    vec_load<NB>(tb, stage_of(1), own);
    auto step = [&](int i, int d) {
        const int K = stage_of(i);
        vec_load<NB>(tb, stage_of(i + 1), nxt);
        double x[1] = {own[0]};
        frag_matvec<NB>(&rm[d], run, x);
        run[0] = x[0];
        vec_store<NB>(tb, K, run, writer);
        own[0] = nxt[0];
    };
    int i0 = 1;
    for (; i0 + DEPTH - 1 <= nsteps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { step(i0 + d, d); ring_load(i0 + d + DEPTH, d); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (i0 + d <= nsteps) step(i0 + d, d);
}

// Tc <- K^-1 Tc through the grouped factor.  Tc: stage-major [N][16]; Tg: [N' + 2][16] right behind it (the two extra slots: the last yh of
// either half, for the middle stage).  All threads call; barriers inside.
__device__ __forceinline__ void kkt_core_group(const CoreArgs &a, double *Tc) {
    constexpr int NB = 16;
    const int g = a.grp, nb = a.nx + a.nu, N = a.N, NS = group_count(N, g), mid = NS / 2;
    const int wv = logical_wave(), lane = opaque_lane(threadIdx.x & 63);
    double *Tg = Tc + N * NB;
    const double *F = a.F;
    for (int idx = threadIdx.x; idx < NS * NB; idx += NT) {
        const int K = idx / NB, A = idx - K * NB, i = A / nb, k = K * g + i;
        Tg[idx] = (A < g * nb && k < N) ? Tc[k * NB + (A - i * nb)] : 0.0;
    }
    __syncthreads();
    TICK(6)
    if (wv == 0) group_fwd_sweep(0, +1, mid - 1, NS, F, Tg);                // w_0 .. w_{mid-1};  yh_{mid-1} -> slot N'
    else if (wv == 1) group_fwd_sweep(NS - 1, -1, NS - 2 - mid, NS + 1, F, Tg);      // w_{N'-1} .. w_{mid+1};  yh_{mid+1} -> slot N'+1
    __syncthreads();
    TICK(1)
    if (wv == 0) {
        double *tb = Tg + vec_lane_offset(lane);
        d4 fu, fd, fs;
        frag_load<NB>(F + (size_t)mid * GroupFmt::REC + GroupFmt::OMH, lane, &fu);
        frag_load<NB>(F + (size_t)NS * GroupFmt::REC + GroupFmt::OMH, lane, &fd);
        frag_load<NB>(F + (size_t)mid * GroupFmt::REC + GroupFmt::OSINV, lane, &fs);
        double up[1], dn[1], acc[1], out[1] = {0.0};
        vec_load<NB>(tb, mid, acc); vec_load<NB>(tb, NS, up); vec_load<NB>(tb, NS + 1, dn);
        frag_matvec<NB>(&fu, up, acc);
        frag_matvec<NB>(&fd, dn, acc);
        frag_matvec<NB>(&fs, acc, out);
        vec_store<NB>(tb, mid, out, vec_lane_writer(lane));
    }
    __syncthreads();
    TICK(2)
    if (wv == 0) group_back_sweep(mid, -1, mid, -1, F, Tg);                  // x_{mid-1} .. x_0:       x_K = w_K - Mh_{K+1}' x_{K+1}
    else if (wv == 1) group_back_sweep(mid, +1, NS - 1 - mid, NS, F, Tg);    // x_{mid+1} .. x_{N'-1}:  x_K = w_K - Mt_{K-1}' x_{K-1}
    __syncthreads();
    TICK(3)
    for (int idx = threadIdx.x; idx < NS * NB; idx += NT) {
        const int K = idx / NB, A = idx - K * NB, i = A / nb, k = K * g + i;
        if (A < g * nb && k < N) Tc[k * NB + (A - i * nb)] = Tg[idx];
    }
    __syncthreads();
}

// Factorization of a handle with grouped stages, the held input's border on top as usual.  (Kept OUT of factor_all: inlined there, the refactorization
// phase of the four-per-CU kernels -- 140 KB of code at 128 registers -- faulted on its first rho update, also for problems that never took this
// branch; on its own, next to it, both are fine.)
template <int NB> __device__ __forceinline__ void border_factor(const Ctx &, const double *, const double *, double, const double *, double *, double *, double *, double *, double *, double *);
__device__ __forceinline__ int factor_grouped(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag, BorderPtrs bp) {
    const int bad = factor_group(c, om, sv, cc, F, W, iflag);
    __syncthreads();
    if (c.L.border) border_factor<16>(c, om, sv, cc, F, bp.Bb, bp.Zb, bp.Sig, W, W + c.L.m, bp.red);
    return bad;
}

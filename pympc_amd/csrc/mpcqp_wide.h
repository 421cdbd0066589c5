// mpcqp_wide.h -- stage blocks wider than 32: 32 < nx + nu <= 64 (the reference accepts any size, mpc.py:90-96).
//
// Such stages are rare and large (a 64 x 64 block is 32 KB), so this backend is plain: a one-directional block LDL' of the same
// block-tridiagonal K the other backends factor (kkt_diag_entry / kkt_sub_entry), Gauss-Jordan inverses in LDS, and mat-vecs on
// the vector ALUs with the whole workgroup on one stage -- a 64 x 64 mat-vec is 16 FMAs per thread, which the 4x4x4 matrix
// instruction (256 of them per mat-vec, 64 per wave) cannot beat.  With M_k = K_{k+1,k} S_k^-1:
//     S_0 = K_00,   S_{k+1} = K_{k+1,k+1} - M_k K_{k+1,k}'
//     forward   y_{k+1} = r_{k+1} - M_k y_k          (N - 1 dependent steps)
//     diagonal  z_k = S_k^-1 y_k                     (independent)
//     backward  x_k = z_k - M_k' x_{k+1}             (N - 1 dependent steps)
// Per stage the factor holds [ S_k^-1 | M_k | M_k' ], each in the order its mat-vec reads it: thread t = 4 r + c owns row r and the
// columns j = 4 q + c, q = 0..15, sixteen consecutive doubles (a wave reads 8 KB contiguous); the quad of a row adds its four
// partial sums by DPP.  The LDS vector reads of a quad are four neighbouring doubles, those of different rows the same address.
#pragma once

struct WideFmt {
    static constexpr int NB = 64, NN = NB * NB;
    static constexpr int STAGE = 3 * NN;                      // [ S^-1 | M | M' ]
    static constexpr int OSINV = 0, OM = NN, OMT = 2 * NN;
    static constexpr int LD = NB + 1;                         // LDS row stride of the factorization's two work matrices
    static constexpr int WS = 2 * NB * LD + NB;               // LDS doubles of the factorization: two work matrices, one stage's dynamics-row weights
};
static_assert(kLatOnly || NT == 4 * WideFmt::NB, "wide stages: four threads per row");
__device__ __forceinline__ int wide_pos(int r, int j) { return (r * 4 + (j & 3)) * 16 + (j >> 2); }

// sum over the four lanes of a quad (DPP quad_perm [1,0,3,2], then [2,3,0,1])
template <int CTRL>
__device__ __forceinline__ double quad_move(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)xi, CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_mov_dpp((int)(xi >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double quad_sum(double x) { x += quad_move<0xB1>(x); return x + quad_move<0x4E>(x); }

struct WideRow { d4 v[4]; };                                  // sixteen matrix entries of one thread
__device__ __forceinline__ WideRow wide_load(const double *M) {
    WideRow w; cgd4 *p = (cgd4 *)(M + (size_t)threadIdx.x * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.v[i] = p[i];
    return w;
}
// this thread's part of one row of  M vec  (vec: LDS, 64 doubles)
__device__ __forceinline__ double wide_dot(const WideRow &w, const double *vec) {
    const double *v = vec + (threadIdx.x & 3);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a0 = fma(w.v[i][0], v[16 * i], a0); a1 = fma(w.v[i][1], v[16 * i + 4], a1);
        a0 = fma(w.v[i][2], v[16 * i + 8], a0); a1 = fma(w.v[i][3], v[16 * i + 12], a1);
    }
    return a0 + a1;
}

// One chain of `count` dependent steps, the matrices of the next DEPTH steps in flight in registers (a lone instance waits a full
// memory round trip per step otherwise: the arithmetic of a step is a few hundred cycles).  mat(i): matrix of step i; step(i, w).
template <int DEPTH, class Mat, class Step>
__device__ __forceinline__ void wide_chain(int count, Mat &&mat, Step &&step) {
    WideRow ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (d < count) ring[d] = wide_load(mat(d));
    for (int i0 = 0; i0 < count; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int i = i0 + d;
            if (i < count) {
                step(i, ring[d]);
                if (i + DEPTH < count) ring[d] = wide_load(mat(i + DEPTH));
            }
        }
    }
}

// Tc <- K^-1 Tc on the padded stage-major vector (stride 64); Zc: LDS, N * 64 doubles.  All threads call; barriers inside.
__device__ __forceinline__ void wide_core(const double *F, int N, double *Tc, double *Zc) {
    constexpr int NB = WideFmt::NB, DEPTH = 6;
    const int tid = threadIdx.x, r = tid >> 2;
    const bool writer = (tid & 3) == 0;
    wide_chain<DEPTH>(N - 1, [&](int k) { return F + (size_t)k * WideFmt::STAGE + WideFmt::OM; }, [&](int k, const WideRow &m) {
        const double t = quad_sum(wide_dot(m, Tc + k * NB));
        if (writer) Tc[(k + 1) * NB + r] -= t;
        __syncthreads();
    });
    wide_chain<DEPTH>(N, [&](int k) { return F + (size_t)k * WideFmt::STAGE + WideFmt::OSINV; }, [&](int k, const WideRow &s) {
        const double t = quad_sum(wide_dot(s, Tc + k * NB));
        if (writer) Zc[k * NB + r] = t;
    });
    __syncthreads();
    if (tid < NB) Tc[(N - 1) * NB + tid] = Zc[(N - 1) * NB + tid];
    __syncthreads();
    wide_chain<DEPTH>(N - 1, [&](int i) { return F + (size_t)(N - 2 - i) * WideFmt::STAGE + WideFmt::OMT; }, [&](int i, const WideRow &m) {
        const int k = N - 2 - i;
        const double t = quad_sum(wide_dot(m, Tc + (k + 1) * NB));
        if (writer) Tc[k * NB + r] = Zc[k * NB + r] - t;
        __syncthreads();
    });
}

// Factorization.  W: LDS, WideFmt::WS doubles.  Thread t = 4 a + c owns the entries (a, 16 c .. 16 c + 15) of every 64 x 64 block.
// Returns 1 on a non-positive pivot.
__device__ __forceinline__ int factor_wide(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag) {
    constexpr int NB = WideFmt::NB, LD = WideFmt::LD, E = 16;
    const Lay &L = c.L;
    const int N = L.N, tid = threadIdx.x, a = tid >> 2, b0 = (tid & 3) * E;
    double *A = W, *Bm = W + NB * LD, *Od = W + 2 * NB * LD;
    if (tid == 0) *iflag = 0;
    // K_kk into A (added to what A holds if `add`).  Its one dense term, G' diag(om_dyn) G with G = [Ad Bd] (nx rows) and om_dyn the weights of
    // the dynamics rows of stage k+1, is formed as a product from LDS -- G zero-padded to 64 columns in Bm, which is free at this point.
    auto diag_block = [&](int k, bool add) {
        const int nbk = (k < L.NcT) ? L.nb : L.nx;
        const bool dyn = k < L.Np;
        if (dyn) {
            if (tid < L.nx) Od[tid] = om[(k + 1) * L.nx + tid];
            for (int e = tid; e < L.nx * NB; e += NT) {
                const int r = e / NB, q = e % NB;
                Bm[r * LD + q] = q < L.nx ? c.Ad()[r * L.nx + q] : (q < nbk ? c.Bd()[r * L.nu + (q - L.nx)] : 0.0);
            }
        }
        __syncthreads();
        double acc[E];
#pragma unroll
        for (int u = 0; u < E; ++u) acc[u] = add ? A[a * LD + b0 + u] : 0.0;
        if (dyn) for (int r = 0; r < L.nx; ++r) {
            const double ga = Bm[r * LD + a] * Od[r];
#pragma unroll
            for (int u = 0; u < E; ++u) acc[u] = fma(ga, Bm[r * LD + b0 + u], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < E; ++u) A[a * LD + b0 + u] = acc[u];
        for (int u = 0; u < E; ++u) A[a * LD + b0 + u] += kkt_diag_entry<false>(c, om, sv, cc, k, a, b0 + u);
        __syncthreads();
    };
    diag_block(0, false);
    for (int k = 0; k < N; ++k) {
        double *Fk = F + (size_t)k * WideFmt::STAGE;
        // A = S_k  ->  S_k^-1, in place (Gauss-Jordan, as in factor_all: step p works from the old pivot row and column).  A thread keeps its
        // sixteen entries in registers through the 64 steps; LDS carries the pivot row and column from their owners to everybody.
        double cur[E];
#pragma unroll
        for (int u = 0; u < E; ++u) cur[u] = A[a * LD + b0 + u];
        for (int pv = 0; pv < NB; ++pv) {
            // (one barrier per step: the steps alternate between A and Bm, which is idle here, and end in A)
            const double *Ar = (pv & 1) ? Bm : A;
            double *Aw = (pv & 1) ? A : Bm;
            double rpj[E];
            double d = Ar[pv * LD + pv];
            const double rip = Ar[a * LD + pv];
#pragma unroll
            for (int u = 0; u < E; ++u) rpj[u] = Ar[pv * LD + b0 + u];
            if (!(d > 0.0)) { if (tid == 0) *iflag = 1; d = 1e-300; }
            const double inv = 1.0 / d, t = rip * inv;
            const bool rowp = a == pv;
#pragma unroll
            for (int u = 0; u < E; ++u) {
                const bool colp = b0 + u == pv;
                const double off = rowp ? rpj[u] * inv : fma(-t, rpj[u], cur[u]), on = rowp ? inv : -t;
                cur[u] = colp ? on : off;
                Aw[a * LD + b0 + u] = cur[u];
            }
            __syncthreads();
        }
        double reg[E];
#pragma unroll
        for (int u = 0; u < E; ++u) reg[u] = 0.5 * (A[a * LD + b0 + u] + A[(b0 + u) * LD + a]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E; ++u) { A[a * LD + b0 + u] = reg[u]; Fk[WideFmt::OSINV + wide_pos(a, b0 + u)] = reg[u]; }
        if (k + 1 == N) break;
        // Bm = K_{k+1,k} (rows: stage k+1);  M = Bm S^-1
        for (int u = 0; u < E; ++u) Bm[a * LD + b0 + u] = kkt_sub_entry(c, om, cc, k, a, b0 + u);
        __syncthreads();
        // The two 64 x 64 x 64 products on the matrix cores (v_mfma_f64_16x16x4_f64, operand map in mpcqp_factor.h): wave w computes the four
        // 16 x 16 tiles of block row w.  The odd row stride makes both access patterns -- down a column for A operands, along a row for B
        // operands -- conflict free.
        const int wv = tid >> 6, lr = tid & 15, lk = (tid >> 4) & 3;
        d4 acc[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[tj] = d4{0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < NB / 4; ++kk) {
            const double av = Bm[(16 * wv + lr) * LD + 4 * kk + lk];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, A[(4 * kk + lk) * LD + 16 * tj + lr], acc[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * wv + lk + 4 * v, col = 16 * tj + lr;
                Fk[WideFmt::OM + wide_pos(row, col)] = acc[tj][v]; Fk[WideFmt::OMT + wide_pos(col, row)] = acc[tj][v];
            }
        __syncthreads();
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) A[(16 * wv + lk + 4 * v) * LD + 16 * tj + lr] = acc[tj][v];      // A = M
        __syncthreads();
        // S_{k+1} = K_{k+1,k+1} - M Bm'
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[tj] = d4{0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < NB / 4; ++kk) {
            const double av = A[(16 * wv + lr) * LD + 4 * kk + lk];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bm[(16 * tj + lr) * LD + 4 * kk + lk], acc[tj], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) A[(16 * wv + lk + 4 * v) * LD + 16 * tj + lr] = -acc[tj][v];
        __syncthreads();
        diag_block(k + 1, true);
    }
    __syncthreads();
    return *iflag;
}

// the two entry points the kernels call, for 64-wide stages
template <>
__device__ __forceinline__ int factor_all<64>(const Ctx &c, const double *om, const double *sv, double cc, double *F, double *W, int *iflag, BorderPtrs bp) {
    const int bad = factor_wide(c, om, sv, cc, F, W, iflag);
    if (c.L.border) border_factor<64>(c, om, sv, cc, F, bp.Bb, bp.Zb, bp.Sig, W, W + c.L.m, bp.red);
    return bad;
}
template <>
__device__ __forceinline__ void kkt_core<64, false>(const CoreArgs &a, double *Tc) { wide_core(a.F, a.N, Tc, Tc + a.N * WideFmt::NB); TICK(3) }

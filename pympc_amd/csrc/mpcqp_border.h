// mpcqp_border.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Control horizon Nc < Np: bordered (Schur complement) correction; generic KKT solve for the tests.
#pragma once

// ------------------------------------------------------------------------------------------------
// Control horizon Nc < Np (mpc.py:513-517,540-543): the last input ub = u_{Nc-1} is held to the end of the
// horizon, so it couples to every later stage and K is block tridiagonal plus a border:
//     K = [ T  B ; B' C ],   T = K without ub (stage Nc-1 keeps only x),   B = K[:, ub],   C = K[ub, ub].
// With Z = T^-1 B and Sigma = C - B'Z (computed at factor time):  ub = Sigma^-1 (r2 - Z' r1),  y = T^-1 (r1 - B ub).
// B and C are taken entry by entry from the matrix-free operators (K = cP + diag(s) + A' diag(omega) A).
// ------------------------------------------------------------------------------------------------
// (ALWAYS inlined.  As an ordinary function it was ONE callee shared by the factor phases of kernels with different launch bounds -- one, two
//  and four workgroups per CU -- and by k_setup; when round 4 added a second caller inside the refactorization phase of the four-per-CU
//  kernels, that phase began to return NaN factors for bordered 16 x 16 problems (or faulted), although no line on its path had changed:
//  scripts/diag_refactor.py.  With internal linkage or inlined the same sources are correct; inlined, no callee is shared at all.)
__device__ __forceinline__ double kkt_entry_generic(const Ctx &c, const double *om, const double *sv, double cc, int v, int w) {
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
__device__ __forceinline__ void border_factor(const Ctx &c, const double *om, const double *sv, double cc, const double *F,
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
        kkt_core<NB>(core_args(L, F, om), Tc);
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

// NU <= 4 inputs: two barriers instead of 4 nu + 2, Sigma^-1 read before the dot products instead of after them (this step was a seventh of an
// iteration of the reference's Kalman notebook alone on its compute unit).  Every thread forms ub itself from the four waves' partial sums; the
// additions are the ones block_reduce makes, in the same order.
template <int NB, int NU>
__device__ __forceinline__ void border_pre_few(const Lay &L, const double *Bb, const double *Zb, const double *Sig, double *Tc, double *ubar, double *red) {
    const int tid = threadIdx.x, NP = L.N * NB;
    const int slot = (L.Nc - 1) * NB + L.nx;
    double sg[NU * NU], part[NU], ub[NU];
#pragma unroll
    for (int e = 0; e < NU * NU; ++e) sg[e] = Sig[e];
#pragma unroll
    for (int j = 0; j < NU; ++j) part[j] = 0.0;
    for (int idx = tid; idx < NP; idx += NT) {
        const double tc = Tc[idx];
#pragma unroll
        for (int j = 0; j < NU; ++j) part[j] += Zb[j * NP + idx] * tc;                  // Z is zero in the r2 slots
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) part[j] = wave_reduce<false>(part[j]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j) red[(tid >> 6) * NU + j] = part[j];
    }
    if (tid < NU) red[NWAVES * NU + tid] = Tc[slot + tid];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        double v = red[j];
#pragma unroll
        for (int w = 1; w < NWAVES; ++w) v += red[w * NU + j];
        part[j] = red[NWAVES * NU + j] - v;                                              // r2 - Z' r1
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) a += sg[i * NU + j] * part[j];
        ub[i] = a;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) if (tid == i) ubar[i] = ub[i];                          // (border_post puts it into the solution)
    for (int idx = tid; idx < NP; idx += NT) {             // (plain loops: with several elements' operands requested together both loops measured slower)
        double a = Tc[idx];
#pragma unroll
        for (int j = 0; j < NU; ++j) a -= Bb[j * NP + idx] * ub[j];
        Tc[idx] = (idx >= slot && idx < slot + NU) ? 0.0 : a;
    }
    __syncthreads();
}
// Before the tridiagonal solve: Tc holds r1 in the padded slots and r2 in the (otherwise padding) u slots of stage
// Nc-1.  Computes ub, leaves it in ubar[] (LDS, nu doubles) and replaces r1 by r1 - B ub.
template <int NB>
__device__ __forceinline__ void border_pre(const Lay &L, const double *Bb, const double *Zb, const double *Sig, double *Tc, double *ubar, double *red) {
    const int tid = threadIdx.x, nu = L.nu, NP = L.N * NB;
    const int slot = (L.Nc - 1) * NB + L.nx;
    switch (nu) {                                          // few inputs (the usual case): compile-time loops, two barriers
        case 1: border_pre_few<NB, 1>(L, Bb, Zb, Sig, Tc, ubar, red); return;
        case 2: border_pre_few<NB, 2>(L, Bb, Zb, Sig, Tc, ubar, red); return;
        case 3: border_pre_few<NB, 3>(L, Bb, Zb, Sig, Tc, ubar, red); return;
        case 4: border_pre_few<NB, 4>(L, Bb, Zb, Sig, Tc, ubar, red); return;
        default: break;
    }
    // (r2 - Z' r1 is formed IN the r2 slots of Tc -- Z is zero there, so the later dot products do not see the change -- and ubar holds nu
    //  doubles only: the 128 doubles of Smem::tv are enough for any nu <= 127)
    for (int j = 0; j < nu; ++j) {
        double vsum[1] = {0.0}, vmax[1] = {0.0};
        for (int idx = tid; idx < NP; idx += NT) vsum[0] += Zb[(size_t)j * NP + idx] * Tc[idx];     // Z is zero in the r2 slots
        block_reduce<1, 1>(vmax, vsum, red);
        if (tid == 0) Tc[slot + j] -= vsum[0];
        __syncthreads();
    }
    if (tid < nu) { double a = 0.0; for (int j = 0; j < nu; ++j) a += Sig[tid * nu + j] * Tc[slot + j]; ubar[tid] = a; }
    __syncthreads();
    for (int idx = tid; idx < NP; idx += NT) {
        double a = Tc[idx];
        for (int j = 0; j < nu; ++j) a -= Bb[(size_t)j * NP + idx] * ubar[j];
        Tc[idx] = (idx >= slot && idx < slot + nu) ? 0.0 : a;      // (the r2 slots are cleared by their own writer: a separate store would race with this one across waves)
    }
    __syncthreads();
}
__device__ __forceinline__ void border_post(const Lay &L, int NB, double *Tc, const double *ubar) {
    if ((int)threadIdx.x < L.nu) Tc[(L.Nc - 1) * NB + L.nx + threadIdx.x] = ubar[threadIdx.x];
    __syncthreads();
}

// Generic front end (verification kernel): flat rhs (global) -> flat solution `out` (global, n doubles).
// Tc: LDS, N*NB doubles.
template <int NB>
__device__ __forceinline__ void kkt_solve(const Ctx &c, const double *om, const double *sv, double cc, const double *F,
                          const double *rg, double *Tc, double *out, BorderPtrs bp, double *ubar) {
    const Lay &L = c.L;
    const double cef = cc * c.eps_feas();
    for (int idx = threadIdx.x; idx < L.N * NB; idx += NT) {
        int k = idx / NB, a = idx % NB;
        double v = 0.0;
        if (a < L.nx) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e];
            double te = 0.0;
            if (L.soft) { te = rg[L.oe + e] / (cef + sv[L.oe + e] + ws); out[L.oe + e] = te; }
            v = rg[e] - ws * te;
        } else if (a < L.nb && k < L.Nc) v = rg[L.ou + k * L.nu + (a - L.nx)];
        Tc[idx] = v;
    }
    __syncthreads();
    const bool bordered = L.border && !L.dense;            // (the dense inverse holds the held input's couplings itself)
    if (bordered) border_pre<NB>(L, bp.Bb, bp.Zb, bp.Sig, Tc, ubar, bp.red);
    if (NB == 16 && L.dense) dense_core<NB>(L, F, Tc, Tc + L.N * NB, dense_slot<NB>(L));
    else if (NB == 16 && L.bcr) {
        for (int idx = L.N * NB + threadIdx.x; idx < L.bcr * NB; idx += NT) Tc[idx] = 0.0;      // (the schedule's padding stages)
        __syncthreads();
        bcr_core_stream(F, L.bcr, Tc, Tc + L.bcr * NB, L.bcrtop);
    }
    else kkt_core<NB>(core_args(L, F, om), Tc);
    if (bordered) border_post(L, NB, Tc, ubar);
    for (int idx = threadIdx.x; idx < L.N * NB; idx += NT) {
        int k = idx / NB, a = idx % NB;
        if (a < L.nx) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e];
            double xe = Tc[idx];
            out[e] = xe;
            if (L.soft) out[L.oe + e] -= (ws / (cef + sv[L.oe + e] + ws)) * xe;
        } else if (a < L.nb && k < L.Nc) out[L.ou + k * L.nu + (a - L.nx)] = Tc[idx];
    }
    __syncthreads();
}

// mpcqp_lat.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The LATENCY round of 16 x 16 stages (MODE_BCR + N: at most one instance per compute unit, BASELINE shape (12, 4, 30)):
// block cyclic reduction with the factor resident in registers (mpcqp_bcr.h), the iterate resident in registers, the
// products with [Ad Bd] on the matrix cores.
//
// What an iteration of one (12,4,30) instance costs on an otherwise idle compute unit with the bandwidth kernels (measured,
// 22 000 cycles): the two chain sweeps 12 700 (30 dependent stage steps), the parallel phases 9 500 -- not work but latency:
// every LDS value another thread produced costs a write, a barrier and a read, and own_rhs / own_update read 60 of them per
// thread one after the other.  Here:
//   * the KKT solve is 2 ceil(log2 N) - 1 level steps of independent mat-vecs; levels 0 and 1 run on all four waves with a
//     barrier each, the levels above them (7 stages, a chain of 5 dependent steps) on ONE wave without barriers;
//     which wave runs which task is fixed at compile time, so each wave keeps exactly the fragments it needs (<= 35 x 8 VGPRs);
//   * every variable and constraint row has an owner thread that keeps it in registers for the round (as mpcqp_tiny.h);
//   * A'W and A x -- the only neighbour couplings -- are 16 x 16 mat-vecs G' W_dyn and G v per stage, G = [Ad Bd]: on the matrix
//     cores with G, G' resident (G v fused into the back substitution: the solution is in the producing wave's registers);
//   * seven barriers per iteration:   E1 + G'W | level 0 | level 1 | upper levels | level 1 back | level 0 back + G v | E2.
#pragma once

// ---- static schedule ---------------------------------------------------------------------------------------------------
// task kinds: 0 kept stage of a forward level (two fragments), 1 eliminated stage of a forward level (D^-1), 2 back substitution
constexpr int lat_count(int N, int L, int kind) { return kind == 0 ? bcr_nk(N, 1 << L) : bcr_ne(N, 1 << L); }
constexpr int lat_stage(int L, int kind, int t) { return kind == 0 ? 2 * (1 << L) * (t + 1) - 1 : (1 << L) * (2 * t + 1) - 1; }
constexpr int lat_nfr(int N, int L, int kind, int t) {
    const int h = 1 << L, s = lat_stage(L, kind, t);
    return kind == 0 ? 1 + (s + h < N ? 1 : 0) : kind == 1 ? 1 : (s - h >= 0 ? 1 : 0) + (s + h < N ? 1 : 0);
}
// which wave runs a task: the levels above 1 all on wave 3 (a dependent chain: one wave, no barriers), which therefore takes
// only the last few forward tasks of levels 0 and 1 and no back-substitution tasks
constexpr int lat_owner(int N, int L, int kind, int t) {
    if (L >= 2) return 3;
    const int n = lat_count(N, L, kind);
    if (kind == 2) return (t + 2 * L) % 3;
    const int to3 = L == 0 ? (kind == 0 ? 2 : 3) : (kind == 0 ? 1 : 2);
    if (t >= n - to3) return 3;
    return (t + kind) % 3;
}
// slot of a task's first fragment in its wave's array: fragments of the wave's earlier tasks (forward levels up, then backward down)
constexpr int lat_slot(int N, int W, int Lq, int kq, int tq) {
    const int LV = bcr_levels(N);
    int s = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int li = 0; li < LV; ++li) {
            const int L = pass == 0 ? li : LV - 1 - li;
            for (int kind = (pass == 0 ? 0 : 2); kind < (pass == 0 ? 2 : 3); ++kind)
                for (int t = 0; t < lat_count(N, L, kind); ++t) {
                    if (L == Lq && kind == kq && t == tq) return s;
                    if (lat_owner(N, L, kind, t) == W) s += lat_nfr(N, L, kind, t);
                }
        }
    return s;
}
constexpr int lat_slots(int N, int W) { return lat_slot(N, W, -1, -1, -1); }
constexpr int lat_max_slots(int N) { int m = 0; for (int w = 0; w < 4; ++w) m = lat_slots(N, w) > m ? lat_slots(N, w) : m; return m; }

#define LAT_LDS_DOUBLES(N) (5 * ((N) + 2) * 16 + NT)      /* LDS doubles of the latency round (five stage-major vectors + Wu) */
#define LAT_DISPATCH(wv, CALL) switch (wv) { \
    case 0: { constexpr int W = 0; CALL; } break; case 1: { constexpr int W = 1; CALL; } break; \
    case 2: { constexpr int W = 2; CALL; } break; default: { constexpr int W = 3; CALL; } break; }

template <int N, int W>
__device__ __forceinline__ void lat_load(const double *F, d4 *fr) {
    const int lane = threadIdx.x & 63;
    constexpr int LV = bcr_levels(N);
    static_for<0, LV>([&](auto lc) {
        constexpr int L = decltype(lc)::value, h = 1 << L;
        static_for<0, lat_count(N, L, 0)>([&](auto tc) {
            constexpr int t = decltype(tc)::value, i = lat_stage(L, 0, t);
            if constexpr (lat_owner(N, L, 0, t) == W) {
                constexpr int s = lat_slot(N, W, L, 0, t);
                fr[s] = bcr_frag(F, i - h, BcrFmt::OLBRT, lane);
                if constexpr (i + h < N) fr[s + 1] = bcr_frag(F, i + h, BcrFmt::OLBLT, lane);
            }
        });
        static_for<0, lat_count(N, L, 1)>([&](auto tc) {
            constexpr int t = decltype(tc)::value, e = lat_stage(L, 1, t);
            if constexpr (lat_owner(N, L, 1, t) == W) { constexpr int s1 = lat_slot(N, W, L, 1, t); fr[s1] = bcr_frag(F, e, BcrFmt::ODINV, lane); }
            if constexpr (lat_owner(N, L, 2, t) == W) {
                constexpr int s = lat_slot(N, W, L, 2, t);
                if constexpr (e - h >= 0) fr[s] = bcr_frag(F, e, BcrFmt::OLBL, lane);
                constexpr int s2 = s + (e - h >= 0 ? 1 : 0);
                if constexpr (e + h < N) fr[s2] = bcr_frag(F, e, BcrFmt::OLBR, lane);
            }
        });
    });
}

// LDS vectors of the round, all stage-major with stride 16 (element a of stage k at k * 16 + a), seen through the per-lane
// base of the MFMA operand layout (vec_lane_offset):
//   tb  right-hand side / solution      cb  c_e of the reduction       ab  G' W_dyn of the next stage (added to the right-hand side)
//   gb  G v of the previous stage (stage k + 1's slot written by the wave that solved stage k)
struct LatVecs { double *tb, *cb, *ab, *gb; };

// ordinal of a task among the tasks of its kind and level that wave W owns (compile time)
constexpr int lat_ord(int N, int W, int L, int kind, int tq) { int o = 0; for (int t = 0; t < tq; ++t) if (lat_owner(N, L, kind, t) == W) ++o; return o; }
constexpr int lat_mine(int N, int W, int L, int kind) { return lat_ord(N, W, L, kind, lat_count(N, L, kind)); }

// (p, q) += A in: the two accumulator chains of a task run through ALL its mat-vecs (one addition per task at the end: a
// double-precision vector add costs as much issue time as an MFMA here, see scripts/diag/mfma_rate.hip)
__device__ __forceinline__ void lat_mv(const d4 a, double in, double &p, double &q) {
    const double r1 = rot_blocks<1>(in), r2 = rot_blocks<2>(in), r3 = rot_blocks<3>(in);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], in, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, q, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, q, 0, 0, 0);
}

// forward tasks of level L that wave W owns.  ADD: the right-hand side is still split in two vectors (tb + ab).
template <int N, int W, int L, bool ADD>
__device__ __forceinline__ void lat_fwd(const d4 *fr, const LatVecs &v) {
    constexpr int h = 1 << L;
    auto rd = [&](int s) { return ADD ? v.tb[s * 16] + v.ab[s * 16] : v.tb[s * 16]; };
    static_for<0, lat_count(N, L, 0)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = lat_stage(L, 0, t);
        if constexpr (lat_owner(N, L, 0, t) == W) {
            constexpr int s = lat_slot(N, W, L, 0, t);
            double p = v.tb[i * 16], q = ADD ? v.ab[i * 16] : 0.0;      // (the stage's own right-hand side starts the two chains)
            lat_mv(fr[s], rd(i - h), p, q);
            if constexpr (i + h < N) lat_mv(fr[s + 1], rd(i + h), p, q);
            v.tb[i * 16] = p + q;
        }
    });
    static_for<0, lat_count(N, L, 1)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 1, t);
        if constexpr (lat_owner(N, L, 1, t) == W) {
            constexpr int s1 = lat_slot(N, W, L, 1, t);
            double p = 0.0, q = 0.0;
            lat_mv(fr[s1], rd(e), p, q);
            v.cb[e * 16] = p + q;
        }
    });
}
// back-substitution tasks of level L that wave W owns; every solved stage e also leaves G v_e for stage e + 1
template <int N, int W, int L>
__device__ __forceinline__ void lat_bwd(const d4 *fr, const d4 Gf, const LatVecs &v) {
    constexpr int h = 1 << L;
    static_for<0, lat_count(N, L, 2)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 2, t);
        if constexpr (lat_owner(N, L, 2, t) == W) {
            constexpr int s = lat_slot(N, W, L, 2, t), s2 = s + (e - h >= 0 ? 1 : 0);
            double p = v.cb[e * 16], q = 0.0;
            if constexpr (e - h >= 0) lat_mv(fr[s], v.tb[(e - h) * 16], p, q);
            if constexpr (e + h < N) lat_mv(fr[s2], v.tb[(e + h) * 16], p, q);
            const double x = p + q;
            v.tb[e * 16] = x;
            if constexpr (e + 1 < N) { double g = 0.0, g2 = 0.0; lat_mv(Gf, x, g, g2); v.gb[(e + 1) * 16] = g + g2; }
        }
    });
}

// Tc <- K^-1 (Tc + AtW), and Gx.  All threads call; six barriers inside, none after the last phase (the caller's follows).
template <int N>
__device__ __forceinline__ void lat_solve(const d4 *fr, const d4 Gf, const LatVecs &v, int wv) {
    constexpr int LV = bcr_levels(N);
    static_assert(LV >= 3, "levels 0 and 1 in parallel, the rest on one wave");
    LAT_DISPATCH(wv, (lat_fwd<N, W, 0, true>(fr, v)))
    __syncthreads();
    TICK(1)
    LAT_DISPATCH(wv, (lat_fwd<N, W, 1, false>(fr, v)))
    __syncthreads();
    TICK(2)
    if (wv == 3) {                 // the upper levels: a chain of dependent level steps, one wave, LDS in program order, no barriers
        static_for<2, LV>([&](auto lc) { lat_fwd<N, 3, decltype(lc)::value, false>(fr, v); });
        // the top level's only stage has no neighbours left: c is the solution
        constexpr int etop = lat_stage(LV - 1, 1, 0);
        const double x = v.cb[etop * 16];
        v.tb[etop * 16] = x;
        if constexpr (etop + 1 < N) { double g = 0.0, g2 = 0.0; lat_mv(Gf, x, g, g2); v.gb[(etop + 1) * 16] = g + g2; }
        static_for<0, LV - 3>([&](auto lc) { lat_bwd<N, 3, LV - 2 - decltype(lc)::value>(fr, Gf, v); });
    }
    __syncthreads();
    TICK(3)
    LAT_DISPATCH(wv, (lat_bwd<N, W, 1>(fr, Gf, v)))
    __syncthreads();
    TICK(4)
    LAT_DISPATCH(wv, (lat_bwd<N, W, 0>(fr, Gf, v)))
}

// ---- the round ---------------------------------------------------------------------------------------------------------
// A constraint row in registers: z, c y / omega, W = omega (z - c y / omega), omega.  Double-precision vector instructions are the
// scarce resource of the element-wise phases (12 cycles of issue each with four waves on the CU: scripts/diag/mfma_rate.hip),
// so a row step is written with as few of them as the arithmetic allows (seven; five for an equality row).
struct LatRow { double z, ys, w, om; };
__device__ __forceinline__ void lat_row_load(LatRow &r, cgdouble *gz, cgdouble *gy, cgdouble *om, double cc, int idx) {
    r.om = om[idx]; r.z = gz[idx]; r.ys = cc * gy[idx] / r.om; r.w = r.om * (r.z - r.ys);
}
// relaxation, projection on [lo, hi], dual step:  zr = alpha zt + beta z;  z+ = clamp(zr + ys);  ys+ = ys + (zr - z+)
// returns the increment of ys (times omega / c: the dual increment the infeasibility certificates look at)
__device__ __forceinline__ double lat_row_step(LatRow &r, double zt, double lo, double hi, double alpha, double beta) {
    const double s = fma(alpha, zt, fma(beta, r.z, r.ys));
    const double zn = fmin(fmax(s, lo), hi);
    const double ysn = s - zn, d = ysn - r.ys;
    r.z = zn; r.ys = ysn; r.w = r.om * (zn - ysn);
    return d;
}
__device__ __forceinline__ double lat_row_step_eq(LatRow &r, double zt, double b0, double alpha, double beta) {
    const double s = fma(alpha, zt, fma(beta, r.z, r.ys));
    const double ysn = s - b0, d = ysn - r.ys;
    r.z = b0; r.ys = ysn; r.w = r.om * (b0 - ysn);
    return d;
}

// fragment of the 16 x 16 matrix whose entry (r, c) is f(r, c), built in registers (operand order: mpcqp_factor.h)
template <class Fn>
__device__ __forceinline__ d4 lat_make_frag(int lane, Fn f) {
    const int k = lane >> 4, b = (lane >> 2) & 3, i = lane & 3, r = 4 * b + i;
    return d4{f(r, 4 * b + k), f(r, 4 * ((b + 1) & 3) + k), f(r, 4 * ((b + 2) & 3) + k), f(r, 4 * ((b + 3) & 3) + k)};
}

template <int NXT, int NUT, int NST>
__device__ __forceinline__ void admm_lat(const Lay &L, const HotPtrs &P, Smem &S, double *Xl, double *Zl, double *Yl, double alpha, int iters) {
    constexpr int NB = 16, nx = NXT, nu = NUT, N = NST, NX = N * nx, NU = (N - 1) * nu;
    static_assert(NX <= 2 * NT && NU <= NT && nx + nu <= NB, "owner map: two state elements and one input element per thread");
    const int b = inst_of(P.perm), tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    gdouble *gx = (gdouble *)(P.x + (size_t)b * L.n), *gz = (gdouble *)(P.z + (size_t)b * L.m), *gy = (gdouble *)(P.y + (size_t)b * L.m);
    cgdouble *om = (cgdouble *)(P.omega + (size_t)b * L.m), *sv = (cgdouble *)(P.s + (size_t)b * L.n), *qv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double cc = P.c[b], cinv = 1.0 / cc, beta = 1.0 - alpha;
    const double *hot = S.hot;
    // LDS: five stage-major vectors of N + 2 stage slots and the flattened Delta-u row vector, in the work area T
    constexpr int VS = (N + 2) * NB;
    double *Tc = S.T, *Cc = Tc + VS, *AtW = Cc + VS, *Gx = AtW + VS, *Wd = Gx + VS, *Wu = Wd + VS;
    for (int i = tid; i < LAT_LDS_DOUBLES(N); i += NT) S.T[i] = 0.0;
    // ---- the factor: this wave's fragments, G = [Ad Bd] (rows: dynamics rows, columns: (x, u)) and G'
    d4 fr[lat_max_slots(N)];
    LAT_DISPATCH(wv, (lat_load<N, W>(P.F + (size_t)b * P.fsz, fr)))
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    auto gent = [&](int r, int c) { return r < nx ? (c < nx ? Ad[r * nx + c] : (c < nx + nu ? Bd[r * nu + (c - nx)] : 0.0)) : 0.0; };
    const d4 Gf = lat_make_frag(lane, gent);
    const d4 GTf = lat_make_frag(lane, [&](int r, int c) { return gent(c, r); });
    const int lo16 = vec_lane_offset(lane);
    const LatVecs vec{Tc + lo16, Cc + lo16, AtW + lo16, Gx + lo16};
    // G' W for FOUR stages per MFMA group: the B operand's four columns j carry four different stages' vectors (the matrix is the
    // same for all of them) -- lane (k, b, j) reads element 4b + k of stage 4g + j + 1, the result for stage 4g + j lands in lane (i, b, j)
    const int lk = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
    const double *wd4 = Wd + (lj + 1) * NB + 4 * lb + lk;      // + 4 g NB
    double *at4 = AtW + lj * NB + 4 * lb + lk;                  // (output: i takes the place of k)
    // ---- owner map (as own_*): state elements e = tid + NT j, input element cu = tid
    const double cef = cc * hot[L.oeps];
    double x[2], ep[2], svx[2], ncq[2], sve[2], kap[2], okap[2], te[2], b0[2];
    LatRow rD[2], rS[2], rI, rU, r0;
    int esl[2], ea[2];
    bool ev[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        ev[j] = e < NX;
        const int ec = ev[j] ? e : 0, k = ec / nx;
        ea[j] = ec - k * nx; esl[j] = k * NB + ea[j];
        x[j] = gx[ec]; svx[j] = sv[ec]; ncq[j] = -cc * qv[ec];
        ep[j] = L.soft ? gx[L.oe + ec] : 0.0; sve[j] = L.soft ? sv[L.oe + ec] : 0.0;
        lat_row_load(rD[j], gz, gy, om, cc, ec);
        lat_row_load(rS[j], gz, gy, om, cc, L.rs + ec);
        kap[j] = L.soft ? 1.0 / (cef + sve[j] + rS[j].om) : 0.0; okap[j] = rS[j].om * kap[j];
        te[j] = 0.0;
        b0[j] = k == 0 ? -S.x0s[ea[j]] : 0.0;
    }
    const bool uv = tid < NU, u0v = tid < nu;
    const int cu = uv ? tid : 0, uk = cu / nu, uj = cu - uk * nu;
    double u = gx[L.ou + cu];
    const double svu = sv[L.ou + cu], ncqu = -cc * qv[L.n_x + cu];
    lat_row_load(rI, gz, gy, om, cc, L.ri + cu);
    lat_row_load(rU, gz, gy, om, cc, L.rdu + nu + cu);
    lat_row_load(r0, gz, gy, om, cc, L.rdu + (u0v ? tid : 0));
    if (!u0v) r0.w = 0.0;
    const int uslot = uk * NB + nx + uj;                                          // this input's slot in the stage-major vectors
    const int unext = (uj + 1 < nu) ? uslot + 1 : (uk + 1) * NB + nx;             // next flattened input (cu + 1 < n_u)
    const bool has_unext = uv && cu + 1 < NU, has_uprev = uv && cu > 0;
    const double s_uprev = has_uprev ? 1.0 : 0.0, s_unext = has_unext ? 1.0 : 0.0;
    const double *uprevp = Wu + (has_uprev ? cu - 1 : 0), *unextp = Tc + (has_unext ? unext : 0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (ev[j]) Wd[esl[j]] = rD[j].w;
    if (uv) Wu[cu] = rU.w;
    __syncthreads();
    TICK_RESET
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;
        TICK_START
        // ---- G' W_dyn of the next stage for every stage: two groups of four stages per wave ...
        {
            double g0 = 0.0, h0 = 0.0, g1 = 0.0, h1 = 0.0;
            lat_mv(GTf, wd4[4 * wv * NB], g0, h0);
            lat_mv(GTf, wd4[4 * (wv + NWAVES) * NB], g1, h1);
            at4[4 * wv * NB] = g0 + h0;
            at4[4 * (wv + NWAVES) * NB] = g1 + h1;
        }
        // ---- ... and the rest of the right-hand side  s x - c q + (own rows' W)  with the slack eliminated (E1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            te[j] = fma(sve[j], ep[j], rS[j].w) * kap[j];                          // (hard state box: kap = 0, no slack)
            const double rhs = (fma(svx[j], x[j], ncq[j]) - rD[j].w) + fma(-rS[j].om, te[j], rS[j].w);
            if (ev[j]) Tc[esl[j]] = rhs;
        }
        {
            const double rhs = fma(s_uprev, *uprevp, fma(svu, u, ncqu) + (rI.w - rU.w)) + r0.w;
            if (uv) Tc[uslot] = rhs;
        }
        __syncthreads();
        TICK(0)
        lat_solve<N>(fr, Gf, vec, wv);
        __syncthreads();
        TICK(5)
        // ---- relaxation, projection, dual step of the owned rows (E2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double xt = Tc[esl[j]], gv = Gx[esl[j]];
            const double xlo = hot[L.oxmin + ea[j]], xhi = hot[L.oxmax + ea[j]];
            const double et = fma(-okap[j], xt, te[j]);
            const double xn = fma(alpha, xt, beta * x[j]), en = fma(alpha, et, beta * ep[j]);
            const double dD = lat_row_step_eq(rD[j], gv - xt, b0[j], alpha, beta);
            const double dS = lat_row_step(rS[j], xt + et, xlo, xhi, alpha, beta);
            if (keep_delta && ev[j]) {
                const int e = tid + NT * j;
                dxg[e] = xn - x[j]; if (L.soft) dxg[L.oe + e] = en - ep[j];
                dyg[e] = (rD[j].om * cinv) * dD; dyg[L.rs + e] = (rS[j].om * cinv) * dS;
            }
            x[j] = xn; ep[j] = en;
            if (ev[j]) Wd[esl[j]] = rD[j].w;
        }
        {
            const double ut = Tc[uslot], un = *unextp;
            const double ulo = hot[L.oumin + uj], uhi = hot[L.oumax + uj], dlo = hot[L.oDumin + uj], dhi = hot[L.oDumax + uj];
            const double unew = fma(alpha, ut, beta * u);
            const double dI = lat_row_step(rI, ut, ulo, uhi, alpha, beta);
            const double dU = lat_row_step(rU, fma(s_unext, un, -ut), dlo, dhi, alpha, beta);
            if (keep_delta && uv) { dxg[L.ou + cu] = unew - u; dyg[L.ri + cu] = (rI.om * cinv) * dI; dyg[L.rdu + nu + cu] = (rU.om * cinv) * dU; }
            u = unew;
            if (u0v) { const double d0 = lat_row_step(r0, ut, S.du0[tid], S.du0[nu + tid], alpha, beta); if (keep_delta) dyg[L.rdu + tid] = (r0.om * cinv) * d0; }
            if (uv) Wu[cu] = rU.w;
        }
        __syncthreads();
        TICK(6)
    }
    TICK_FLUSH
    // ---- end of the round: the iterate back to memory (global: next round / warm start; LDS copy: the residual evaluation)
    auto put_row = [&](const LatRow &r, int idx) { const double y = r.ys * (r.om * cinv); gz[idx] = r.z; gy[idx] = y; Zl[idx] = r.z; Yl[idx] = y; };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (ev[j]) {
            gx[e] = x[j]; Xl[e] = x[j];
            if (L.soft) { gx[L.oe + e] = ep[j]; Xl[L.oe + e] = ep[j]; }
            put_row(rD[j], e); put_row(rS[j], L.rs + e);
        }
    }
    if (uv) { gx[L.ou + cu] = u; Xl[L.ou + cu] = u; put_row(rI, L.ri + cu); put_row(rU, L.rdu + nu + cu); }
    if (u0v) put_row(r0, L.rdu + tid);
}

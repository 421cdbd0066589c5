// mpcqp_lat.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The LATENCY round of 16 x 16 stages on 256-thread workgroups (MODE_BCR + N: forced only since round 5 -- AUTO runs mpcqp_latw.h; any nx + nu <= 16 and up to 31 stages, the BASELINE
// shape (12, 4, 30) with compile-time dimensions):
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
//   * every variable and constraint row has an owner lane that keeps it in registers for the round (as mpcqp_tiny.h), and the
//     owner map IS the MFMA operand layout: lane (I, B, J) of group g owns slot a = 4B + I of stage s = 4g + J (x_s[a] with its
//     slack, dynamics row and box row, or u_s[a - nx] with its box row and Delta-u row), two groups per wave;
//   * A'W and A x -- the only neighbour couplings -- are 16 x 16 mat-vecs G' W_dyn(s+1) and G v(s-1), G = [Ad Bd]; the same G for
//     every stage, so ONE MFMA group does FOUR stages (the B operand's four columns carry four stages' vectors) and its result
//     lands exactly in the owner lanes: no LDS round trip between the products and the rows that consume them;
//   * seven barriers per iteration:  G v + row updates | G'W + right-hand side | level 0 | level 1 | upper levels | level 1 back | level 0 back.
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

#define LAT_VS(N) ((4 * (((N) + 3) / 4) + 2) * 16)          /* doubles of one stage-major vector: stage slots -1 .. 4 ceil(N/4) */
#define LAT_LDS_DOUBLES(N) (4 * LAT_VS(N) + 16)              /* LDS doubles of the latency round: right-hand side, c_e, W of the slots' two rows */
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
//   tb  right-hand side / solution      cb  c_e of the reduction
//   t1, t2, t3: tb seen through the lane bases of the three block rotations -- an input vector is read FOUR times from LDS
//   (one 8-byte read per MFMA step) instead of once plus six cross-lane moves on the vector ALU, which is the busier unit here
struct LatVecs { double *tb, *cb; const double *t1, *t2, *t3; };
__device__ __forceinline__ void lat_mv_lds(const d4 a, const LatVecs &v, int off, double &p, double &q) {
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], v.tb[off], p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], v.t2[off], q, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], v.t1[off], p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], v.t3[off], q, 0, 0, 0);
}

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

// forward tasks of level L that wave W owns
template <int N, int W, int L>
__device__ __forceinline__ void lat_fwd(const d4 *fr, const LatVecs &v) {
    constexpr int h = 1 << L;
    static_for<0, lat_count(N, L, 0)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = lat_stage(L, 0, t);
        if constexpr (lat_owner(N, L, 0, t) == W) {
            constexpr int s = lat_slot(N, W, L, 0, t);
            double p = v.tb[i * 16], q = 0.0;                 // (the stage's own right-hand side starts the chain)
            lat_mv_lds(fr[s], v, (i - h) * 16, p, q);
            if constexpr (i + h < N) lat_mv_lds(fr[s + 1], v, (i + h) * 16, p, q);
            v.tb[i * 16] = p + q;
        }
    });
    static_for<0, lat_count(N, L, 1)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 1, t);
        if constexpr (lat_owner(N, L, 1, t) == W) {
            constexpr int s1 = lat_slot(N, W, L, 1, t);
            double p = 0.0, q = 0.0;
            lat_mv_lds(fr[s1], v, e * 16, p, q);
            v.cb[e * 16] = p + q;
        }
    });
}
// back-substitution tasks of level L that wave W owns
template <int N, int W, int L>
__device__ __forceinline__ void lat_bwd(const d4 *fr, const LatVecs &v) {
    constexpr int h = 1 << L;
    static_for<0, lat_count(N, L, 2)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 2, t);
        if constexpr (lat_owner(N, L, 2, t) == W) {
            constexpr int s = lat_slot(N, W, L, 2, t), s2 = s + (e - h >= 0 ? 1 : 0);
            double p = v.cb[e * 16], q = 0.0;
            if constexpr (e - h >= 0) lat_mv_lds(fr[s], v, (e - h) * 16, p, q);
            if constexpr (e + h < N) lat_mv_lds(fr[s2], v, (e + h) * 16, p, q);
            v.tb[e * 16] = p + q;
        }
    });
}

// Tc <- K^-1 Tc.  All threads call; four barriers inside, none after the last phase (the caller's follows).
template <int N>
__device__ __forceinline__ void lat_solve(const d4 *fr, const LatVecs &v, int wv) {
    constexpr int LV = bcr_levels(N);
    static_assert(LV >= 3, "levels 0 and 1 in parallel, the rest on one wave");
    LAT_DISPATCH(wv, (lat_fwd<N, W, 0>(fr, v)))
    __syncthreads();
    TICK(1)
    LAT_DISPATCH(wv, (lat_fwd<N, W, 1>(fr, v)))
    __syncthreads();
    TICK(2)
    if (wv == 3) {                 // the upper levels: a chain of dependent level steps, one wave, LDS in program order, no barriers
        static_for<2, LV>([&](auto lc) { lat_fwd<N, 3, decltype(lc)::value>(fr, v); });
        constexpr int etop = lat_stage(LV - 1, 1, 0);        // the top level's only stage has no neighbours left: c is the solution
        v.tb[etop * 16] = v.cb[etop * 16];
        static_for<0, LV - 3>([&](auto lc) { lat_bwd<N, 3, LV - 2 - decltype(lc)::value>(fr, v); });
    }
    __syncthreads();
    TICK(3)
    LAT_DISPATCH(wv, (lat_bwd<N, W, 1>(fr, v)))
    __syncthreads();
    TICK(4)
    LAT_DISPATCH(wv, (lat_bwd<N, W, 0>(fr, v)))
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
    // NST: stage count of the static schedule (compile time), NR <= NST the problem's own (run time); NXT / NUT = 0: nx, nu from the layout
    constexpr int NB = 16, N = NST, NG = (N + 3) / 4;
    static_assert(NG <= 2 * NWAVES && NXT + NUT <= NB, "owner map: two groups of four stages per wave");
    const int nx = NXT ? NXT : L.nx, nu = NUT ? NUT : L.nu, NR = L.N;
    const int b = inst_of(P.perm), tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    gdouble *gx = (gdouble *)(P.x + (size_t)b * L.n), *gz = (gdouble *)(P.z + (size_t)b * L.m), *gy = (gdouble *)(P.y + (size_t)b * L.m);
    cgdouble *om = (cgdouble *)(P.omega + (size_t)b * L.m), *sv = (cgdouble *)(P.s + (size_t)b * L.n), *qv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double cc = P.c[b], cinv = 1.0 / cc, beta = 1.0 - alpha;
    const double *hot = S.hot;
    // LDS: four stage-major vectors with stage slots -1 .. 4 NG (the slots outside 0 .. N-1 stay zero: neighbours of the ends)
    constexpr int VS = LAT_VS(N);
    double *Tc = S.T + NB, *Cc = Tc + VS, *WA = Cc + VS, *WB = WA + VS;      // right-hand side / solution, c_e, W of the first / second row of a slot
    TICK_RESET
    TICK_START
    for (int i = tid; i < LAT_LDS_DOUBLES(N); i += NT) S.T[i] = 0.0;
    // ---- the factor: this wave's fragments, G = [Ad Bd] (rows: dynamics rows, columns: (x, u)) and G'
    d4 fr[lat_max_slots(N)];
    LAT_DISPATCH(wv, (lat_load<N, W>(P.F + (size_t)b * P.fsz, fr)))
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    auto gent = [&](int r, int c) { return r < nx ? (c < nx ? Ad[r * nx + c] : (c < nx + nu ? Bd[r * nu + (c - nx)] : 0.0)) : 0.0; };
    const d4 Gf = lat_make_frag(lane, gent);
    const d4 GTf = lat_make_frag(lane, [&](int r, int c) { return gent(c, r); });
    TICK(7)
    const int lo16 = vec_lane_offset(lane);
    const int lI = lane >> 4, lB = (lane >> 2) & 3;
    const LatVecs vec{Tc + lo16, Cc + lo16, Tc + 4 * ((lB + 1) & 3) + lI, Tc + 4 * ((lB + 2) & 3) + lI, Tc + 4 * ((lB + 3) & 3) + lI};
    // ---- owner map: lane (I, B, J), group g = wv + 4 q  ->  slot a = 4B + I of stage s = 4g + J
    const int a = 4 * ((lane >> 2) & 3) + (lane >> 4), J = lane & 3;
    const bool is_x = a < nx;
    const int jj = a - nx;
    const double cef = cc * hot[L.oeps];
    double pv[2], pv2[2], svp[2], ncq[2], sve[2], kap[2], okap[2], te[2], loA[2], hiA[2], loB[2], hiB[2];
    LatRow rA[2], rB[2], r0{0.0, 0.0, 0.0, 1.0};
    int sl[2], pidx[2], aidx[2], bidx[2];
    bool ok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        // (a group beyond the schedule's last -- shorter schedules have fewer than eight -- owns nothing: it is parked on the zero slot behind the last stage)
        const int g = wv + NWAVES * q, s = g < NG ? 4 * g + J : 4 * NG;
        sl[q] = s * NB + a;
        ok[q] = is_x ? s < NR : (a < nx + nu && s < NR - 1);
        const int e = s * nx + a, cu = s * nu + jj;
        pidx[q] = is_x ? e : L.ou + cu; aidx[q] = is_x ? e : L.ri + cu; bidx[q] = is_x ? L.rs + e : L.rdu + nu + cu;
        pv[q] = pv2[q] = svp[q] = ncq[q] = sve[q] = kap[q] = okap[q] = te[q] = 0.0;
        loA[q] = hiA[q] = loB[q] = hiB[q] = 0.0;
        rA[q] = LatRow{0.0, 0.0, 0.0, 1.0}; rB[q] = LatRow{0.0, 0.0, 0.0, 1.0};
        if (ok[q]) {
            pv[q] = gx[pidx[q]]; svp[q] = sv[pidx[q]]; ncq[q] = -cc * qv[is_x ? e : L.n_x + cu];
            lat_row_load(rA[q], gz, gy, om, cc, aidx[q]);
            lat_row_load(rB[q], gz, gy, om, cc, bidx[q]);
            if (is_x) {
                if (L.soft) { pv2[q] = gx[L.oe + e]; sve[q] = sv[L.oe + e]; kap[q] = 1.0 / (cef + sve[q] + rB[q].om); okap[q] = rB[q].om * kap[q]; }
                loA[q] = hiA[q] = s == 0 ? -S.x0s[a] : 0.0;
                loB[q] = hot[L.oxmin + a]; hiB[q] = hot[L.oxmax + a];
            } else {
                loA[q] = hot[L.oumin + jj]; hiA[q] = hot[L.oumax + jj]; loB[q] = hot[L.oDumin + jj]; hiB[q] = hot[L.oDumax + jj];
            }
        }
    }
    const bool u0v = wv == 0 && J == 0 && !is_x && a < nx + nu;      // the first-step rows u_0 - u_{-1} (mpc.py:574): stage 0's inputs
    if (u0v) lat_row_load(r0, gz, gy, om, cc, L.rdu + jj);
    // neighbours in the flattened input sequence (mpc.py:570): slot a + 1 / a - 1, across the stage boundary at the ends
    const int onext = (jj + 1 < nu) ? 1 : NB - nu + 1, oprev = (jj > 0) ? -1 : -(NB - nu + 1);
    const double selx = is_x ? 1.0 : 0.0, selu = is_x ? 0.0 : 1.0, sgnA = is_x ? -1.0 : 1.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) { WA[sl[q]] = rA[q].w; WB[sl[q]] = rB[q].w; }
    __syncthreads();
    TICK(8)
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;
        TICK_START
        // ---- right-hand side  s x - c q + A'W  with the slack eliminated; A'W = own rows' W + G' W_dyn of the next stage (MFMA, four stages a group)
        {
            double g[2] = {0.0, 0.0}, h2[2] = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 2; ++q) lat_mv(GTf, WA[sl[q] + NB], g[q], h2[q]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double wprev = WB[sl[q] + oprev];
                te[q] = fma(sve[q], pv2[q], rB[q].w) * kap[q];                     // (inputs and a hard state box: kap = 0, no slack)
                const double viaB = is_x ? fma(-rB[q].om, te[q], rB[q].w) : wprev - rB[q].w;
                double rhs = fma(svp[q], pv[q], ncq[q]) + (g[q] + h2[q]);
                rhs = fma(sgnA, rA[q].w, rhs) + viaB;
                if (q == 0) rhs += r0.w;
                Tc[sl[q]] = ok[q] ? rhs : 0.0;
            }
        }
        __syncthreads();
        TICK(0)
        lat_solve<N>(fr, vec, wv);
        __syncthreads();
        TICK(5)
        // ---- G v of the previous stage (MFMA), relaxation, projection, dual step of the owned rows
        {
            double g[2] = {0.0, 0.0}, h2[2] = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 2; ++q) lat_mv(Gf, Tc[sl[q] - NB], g[q], h2[q]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double xt = Tc[sl[q]], un = Tc[sl[q] + onext];
                const double et = fma(-okap[q], xt, te[q]);
                const double vn = fma(alpha, xt, beta * pv[q]), en = fma(alpha, et, beta * pv2[q]);
                const double ztA = fma(selx, (g[q] + h2[q]) - xt - xt, xt);       // x: G v - xt;  u: ut
                const double ztB = is_x ? xt + et : un - xt;
                const double dA = lat_row_step(rA[q], ztA, loA[q], hiA[q], alpha, beta);
                const double dB = lat_row_step(rB[q], ztB, loB[q], hiB[q], alpha, beta);
                if (keep_delta && ok[q]) {
                    dxg[pidx[q]] = vn - pv[q];
                    if (is_x && L.soft) dxg[pidx[q] + L.oe] = en - pv2[q];
                    dyg[aidx[q]] = (rA[q].om * cinv) * dA; dyg[bidx[q]] = (rB[q].om * cinv) * dB;
                }
                pv[q] = vn; pv2[q] = en;
                if (q == 0 && u0v) { const double d0 = lat_row_step(r0, xt, S.du0[jj], S.du0[nu + jj], alpha, beta); if (keep_delta) dyg[L.rdu + jj] = (r0.om * cinv) * d0; }
                WA[sl[q]] = ok[q] ? rA[q].w : 0.0; WB[sl[q]] = ok[q] ? rB[q].w : 0.0;      // (a slot without a variable -- beyond the last stage, the last stage's inputs -- contributes nothing)
            }
        }
        __syncthreads();
        TICK(6)
    }
    TICK_START
    // ---- end of the round: the iterate back to memory (global: next round / warm start; LDS copy: the residual evaluation)
    auto put_row = [&](const LatRow &r, int idx) { const double y = r.ys * (r.om * cinv); gz[idx] = r.z; gy[idx] = y; Zl[idx] = r.z; Yl[idx] = y; };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (ok[q]) {
            gx[pidx[q]] = pv[q]; Xl[pidx[q]] = pv[q];
            if (is_x && L.soft) { gx[pidx[q] + L.oe] = pv2[q]; Xl[pidx[q] + L.oe] = pv2[q]; }
            put_row(rA[q], aidx[q]); put_row(rB[q], bidx[q]);
        }
    }
    if (u0v) put_row(r0, L.rdu + jj);
    TICK(9)
    TICK_FLUSH
    (void)selu;
}

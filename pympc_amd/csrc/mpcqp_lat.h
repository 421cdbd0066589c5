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

// A wave issues in order: written task by task, every task would wait for its own LDS reads, rotations and dependent MFMA pair
// (~350 cycles) before the next one starts.  The tasks of a phase are therefore executed in LOCKSTEP: all LDS reads, fence, all
// rotations, the first MFMA of every chain, the second of every chain, sums and stores -- the matrix pipe stays busy and the
// latencies are paid once per phase, not once per task.
struct LatMv { double in, p, q; };                            // one mat-vec in flight: input vector, the two accumulator chains
// NM mat-vecs in lockstep: m[o].(p, q) += frag(o) m[o].in.  The rotations are made where they are used (a rotated copy lives for one MFMA).
template <int NM, class FragOf>
__device__ __forceinline__ void lat_group(LatMv *m, FragOf frag) {
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        const d4 a = frag(oc);
        m[o].p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], m[o].in, m[o].p, 0, 0, 0);
        m[o].q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], rot_blocks<2>(m[o].in), m[o].q, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto oc) {
        constexpr int o = decltype(oc)::value;
        const d4 a = frag(oc);
        m[o].p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], rot_blocks<1>(m[o].in), m[o].p, 0, 0, 0);
        m[o].q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], rot_blocks<3>(m[o].in), m[o].q, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}
// t-th task (of a kind, at a level) that wave W owns -> its index among all tasks of that kind and level
constexpr int lat_nth(int N, int W, int L, int kind, int o) { int c = 0; for (int t = 0; t < lat_count(N, L, kind); ++t) if (lat_owner(N, L, kind, t) == W) { if (c == o) return t; ++c; } return 0; }

// forward tasks of level L that wave W owns.  ADD: the right-hand side is still split in two vectors (tb + ab).
template <int N, int W, int L, bool ADD>
__device__ __forceinline__ void lat_fwd(const d4 *fr, const LatVecs &v) {
    constexpr int h = 1 << L, NK = lat_mine(N, W, L, 0), NE = lat_mine(N, W, L, 1);
    auto rd = [&](int s) { return ADD ? v.tb[s * 16] + v.ab[s * 16] : v.tb[s * 16]; };
    if constexpr (NK > 0) {
        LatMv ka[NK], kb[NK];
        static_for<0, NK>([&](auto oc) {
            constexpr int o = decltype(oc)::value, i = lat_stage(L, 0, lat_nth(N, W, L, 0, o));
            ka[o].p = rd(i); ka[o].q = 0.0; ka[o].in = rd(i - h);
            kb[o].p = 0.0; kb[o].q = 0.0; kb[o].in = rd(i + h < N ? i + h : i);
        });
        lat_group<NK>(ka, [&](auto oc) { constexpr int s = lat_slot(N, W, L, 0, lat_nth(N, W, L, 0, decltype(oc)::value)); return fr[s]; });
        lat_group<NK>(kb, [&](auto oc) {
            constexpr int t = lat_nth(N, W, L, 0, decltype(oc)::value), s = lat_slot(N, W, L, 0, t);
            if constexpr (lat_stage(L, 0, t) + h < N) return fr[s + 1]; else return d4{0.0, 0.0, 0.0, 0.0};
        });
        static_for<0, NK>([&](auto oc) {
            constexpr int o = decltype(oc)::value, i = lat_stage(L, 0, lat_nth(N, W, L, 0, o));
            v.tb[i * 16] = (ka[o].p + ka[o].q) + (kb[o].p + kb[o].q);
        });
    }
    if constexpr (NE > 0) {
        LatMv ce[NE];
        static_for<0, NE>([&](auto oc) {
            constexpr int o = decltype(oc)::value, e = lat_stage(L, 1, lat_nth(N, W, L, 1, o));
            ce[o].p = 0.0; ce[o].q = 0.0; ce[o].in = rd(e);
        });
        lat_group<NE>(ce, [&](auto oc) { constexpr int s = lat_slot(N, W, L, 1, lat_nth(N, W, L, 1, decltype(oc)::value)); return fr[s]; });
        static_for<0, NE>([&](auto oc) {
            constexpr int o = decltype(oc)::value, e = lat_stage(L, 1, lat_nth(N, W, L, 1, o));
            v.cb[e * 16] = ce[o].p + ce[o].q;
        });
    }
}
// back-substitution tasks of level L that wave W owns; every solved stage e also leaves G v_e for stage e + 1
template <int N, int W, int L>
__device__ __forceinline__ void lat_bwd(const d4 *fr, const d4 Gf, const LatVecs &v) {
    constexpr int h = 1 << L, NE = lat_mine(N, W, L, 2);
    if constexpr (NE > 0) {
        LatMv ma[NE], mb[NE];
        static_for<0, NE>([&](auto oc) {
            constexpr int o = decltype(oc)::value, e = lat_stage(L, 2, lat_nth(N, W, L, 2, o));
            ma[o].p = v.cb[e * 16]; ma[o].q = 0.0; ma[o].in = v.tb[(e - h >= 0 ? e - h : e) * 16];
            mb[o].p = 0.0; mb[o].q = 0.0; mb[o].in = v.tb[(e + h < N ? e + h : e) * 16];
        });
        lat_group<NE>(ma, [&](auto oc) {
            constexpr int t = lat_nth(N, W, L, 2, decltype(oc)::value), s = lat_slot(N, W, L, 2, t);
            if constexpr (lat_stage(L, 2, t) - h >= 0) return fr[s]; else return d4{0.0, 0.0, 0.0, 0.0};
        });
        lat_group<NE>(mb, [&](auto oc) {
            constexpr int t = lat_nth(N, W, L, 2, decltype(oc)::value), e = lat_stage(L, 2, t), s = lat_slot(N, W, L, 2, t) + (e - h >= 0 ? 1 : 0);
            if constexpr (e + h < N) return fr[s]; else return d4{0.0, 0.0, 0.0, 0.0};
        });
        static_for<0, NE>([&](auto oc) {
            constexpr int o = decltype(oc)::value, e = lat_stage(L, 2, lat_nth(N, W, L, 2, o));
            const double x = (ma[o].p + ma[o].q) + (mb[o].p + mb[o].q);
            v.tb[e * 16] = x;
            ma[o].in = x; ma[o].p = 0.0; ma[o].q = 0.0;
        });
        lat_group<NE>(ma, [&](auto) { return Gf; });
        static_for<0, NE>([&](auto oc) {
            constexpr int o = decltype(oc)::value, e = lat_stage(L, 2, lat_nth(N, W, L, 2, o));
            if constexpr (e + 1 < N) v.gb[(e + 1) * 16] = ma[o].p + ma[o].q;
        });
    }
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
        if constexpr (etop + 1 < N) { double g = 0.0, g2 = 0.0; bcr_mv(Gf, x, g, g2); v.gb[(etop + 1) * 16] = g + g2; }
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
struct LatRow { double z, ys, om; };                          // a constraint row in registers: z, c y / omega, omega  (W = omega (z - ys))
__device__ __forceinline__ void lat_row_load(LatRow &r, cgdouble *gz, cgdouble *gy, cgdouble *om, double cc, int idx) {
    r.om = om[idx]; r.z = gz[idx]; r.ys = cc * gy[idx] / r.om;
}
__device__ __forceinline__ double lat_row_w(const LatRow &r) { return r.om * (r.z - r.ys); }
// relaxation, projection on [lo, hi], dual step (own_update's `row`); returns the dual increment in y units
__device__ __forceinline__ double lat_row_step(LatRow &r, double zt, double lo, double hi, double alpha, double beta, double cinv) {
    lo = lo < -QP_INFTY ? -QP_INFTY : lo; hi = hi > QP_INFTY ? QP_INFTY : hi;
    const double zr = alpha * zt + beta * r.z;
    const double zn = fmin(fmax(zr + r.ys, lo), hi);
    const double d = zr - zn;
    r.ys += d; r.z = zn;
    return (r.om * cinv) * d;
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
    // LDS: six stage-major vectors of (N + 1) stages and the flattened Delta-u row vector, in the work area T
    constexpr int VS = (N + 1) * NB;
    double *Tc = S.T, *Cc = Tc + VS, *AtW = Cc + VS, *Gx = AtW + VS, *Wd = Gx + VS, *Wu = Wd + VS;
    for (int i = tid; i < 5 * VS + NT; i += NT) S.T[i] = 0.0;
    // ---- the factor: this wave's fragments, G = [Ad Bd] (rows: dynamics rows, columns: (x, u)) and G'
    d4 fr[lat_max_slots(N)];
    LAT_DISPATCH(wv, (lat_load<N, W>(P.F + (size_t)b * P.fsz, fr)))
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    auto gent = [&](int r, int c) { return r < nx ? (c < nx ? Ad[r * nx + c] : (c < nx + nu ? Bd[r * nu + (c - nx)] : 0.0)) : 0.0; };
    const d4 Gf = lat_make_frag(lane, gent);
    const d4 GTf = lat_make_frag(lane, [&](int r, int c) { return gent(c, r); });
    const int lo16 = vec_lane_offset(lane);
    const LatVecs vec{Tc + lo16, Cc + lo16, AtW + lo16, Gx + lo16};
    const double *wdb = Wd + lo16;
    // ---- owner map (as own_*): state elements e = tid + NT j, input element cu = tid
    const double cef = cc * hot[L.oeps];
    double x[2], ep[2], svx[2], cqx[2], sve[2], kap[2], te[2];
    LatRow rD[2], rS[2], rI, rU, r0;
    int ek[2], ea[2];
    bool ev[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        ev[j] = e < NX;
        const int ec = ev[j] ? e : 0;
        ek[j] = ec / nx; ea[j] = ec - ek[j] * nx;
        x[j] = gx[ec]; svx[j] = sv[ec]; cqx[j] = cc * qv[ec];
        ep[j] = L.soft ? gx[L.oe + ec] : 0.0; sve[j] = L.soft ? sv[L.oe + ec] : 0.0;
        lat_row_load(rD[j], gz, gy, om, cc, ec);
        lat_row_load(rS[j], gz, gy, om, cc, L.rs + ec);
        kap[j] = 1.0 / (cef + sve[j] + rS[j].om);
        te[j] = 0.0;
    }
    const bool uv = tid < NU, u0v = tid < nu;
    const int cu = uv ? tid : 0, uk = cu / nu, uj = cu - uk * nu;
    double u = gx[L.ou + cu];
    const double svu = sv[L.ou + cu], cqu = cc * qv[L.n_x + cu];
    lat_row_load(rI, gz, gy, om, cc, L.ri + cu);
    lat_row_load(rU, gz, gy, om, cc, L.rdu + nu + cu);
    lat_row_load(r0, gz, gy, om, cc, L.rdu + (u0v ? tid : 0));
    const int uslot = uk * NB + nx + uj;                                          // this input's slot in the stage-major vectors
    const int unext = (uj + 1 < nu) ? uslot + 1 : (uk + 1) * NB + nx;             // next flattened input (cu + 1 < n_u)
    const bool has_unext = uv && cu + 1 < NU, has_uprev = uv && cu > 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (ev[j]) Wd[ek[j] * NB + ea[j]] = lat_row_w(rD[j]);
    if (uv) Wu[cu] = lat_row_w(rU);
    __syncthreads();
    TICK_RESET
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;
        TICK_START
        // ---- G' W_dyn of the next stage, for every stage but the last (30 independent mat-vecs over four waves) ...
        {
            constexpr int NM = (N - 1 + NWAVES - 1) / NWAVES;
            LatMv mm[NM];
            static_for<0, NM>([&](auto uc) {
                constexpr int uu = decltype(uc)::value;
                const int k = min(wv + NWAVES * uu, N - 2);   // (a wave short of a task repeats the last one: same value, same place)
                mm[uu].in = wdb[(k + 1) * NB]; mm[uu].p = 0.0; mm[uu].q = 0.0;
            });
            lat_group<NM>(mm, [&](auto) { return GTf; });
            static_for<0, NM>([&](auto uc) {
                constexpr int uu = decltype(uc)::value;
                const int k = min(wv + NWAVES * uu, N - 2);
                vec.ab[k * NB] = mm[uu].p + mm[uu].q;
            });
        }
        // ---- ... and the rest of the right-hand side  s x - c q + (own rows' W)  with the slack eliminated (E1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double wS = lat_row_w(rS[j]);
            te[j] = L.soft ? (sve[j] * ep[j] + wS) * kap[j] : 0.0;
            const double rhs = svx[j] * x[j] - cqx[j] - lat_row_w(rD[j]) + (wS - rS[j].om * te[j]);
            if (ev[j]) Tc[ek[j] * NB + ea[j]] = rhs;
        }
        {
            double rhs = svu * u - cqu + lat_row_w(rI) - lat_row_w(rU);
            if (u0v) rhs += lat_row_w(r0);
            if (has_uprev) rhs += Wu[cu - 1];
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
            const int sl = ek[j] * NB + ea[j];
            const double xt = Tc[sl], gv = Gx[sl];
            const double et = L.soft ? te[j] - (rS[j].om * kap[j]) * xt : 0.0;
            const double xn = alpha * xt + beta * x[j], en = alpha * et + beta * ep[j];
            const int e = tid + NT * j;
            if (keep_delta && ev[j]) { dxg[e] = xn - x[j]; if (L.soft) dxg[L.oe + e] = en - ep[j]; }
            x[j] = xn; ep[j] = en;
            const double b0 = ek[j] == 0 ? -S.x0s[ea[j]] : 0.0;
            const double dD = lat_row_step(rD[j], gv - xt, b0, b0, alpha, beta, cinv);
            const double dS = lat_row_step(rS[j], xt + et, hot[L.oxmin + ea[j]], hot[L.oxmax + ea[j]], alpha, beta, cinv);
            if (keep_delta && ev[j]) { dyg[e] = dD; dyg[L.rs + e] = dS; }
            if (ev[j]) Wd[sl] = lat_row_w(rD[j]);
        }
        {
            const double ut = Tc[uslot], un = has_unext ? Tc[unext] : 0.0;
            const double unew = alpha * ut + beta * u;
            if (keep_delta && uv) dxg[L.ou + cu] = unew - u;
            u = unew;
            const double dI = lat_row_step(rI, ut, hot[L.oumin + uj], hot[L.oumax + uj], alpha, beta, cinv);
            const double dU = lat_row_step(rU, (has_unext ? un : 0.0) - ut, hot[L.oDumin + uj], hot[L.oDumax + uj], alpha, beta, cinv);
            if (keep_delta && uv) { dyg[L.ri + cu] = dI; dyg[L.rdu + nu + cu] = dU; }
            if (u0v) { const double d0 = lat_row_step(r0, ut, S.du0[tid], S.du0[nu + tid], alpha, beta, cinv); if (keep_delta) dyg[L.rdu + tid] = d0; }
            if (uv) Wu[cu] = lat_row_w(rU);
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

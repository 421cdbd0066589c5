// mpcqp_tiny.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The ADMM round of SMALL problems (MODE_DENSE: N (nx+nu) <= 128, the reference's own examples): iterate in REGISTERS.
//
// One small QP alone on a compute unit is bound by latency, not by work: every LDS access that another thread's result depends
// on costs a write, a barrier and a read (~350 cycles with one wave per SIMD), every dependent LDS read ~130.  The generic
// parallel phases (own_rhs / own_update) keep the iterate in LDS and take ~16 such round trips per iteration (6 000 cycles for a
// cart-pole QP of 188 variables).  Here every variable has an OWNER LANE that keeps everything about it in registers for the whole
// round -- the variable, its slack, its rows' z and (scaled) y, the metric values, its column of Ad (or Bd) and its row of
// [Ad Bd] -- and an iteration exchanges exactly three vectors through LDS, one barrier each:
//     (A) owners -> compact right-hand side            | barrier |  every lane pair: row of K^-1 times it (registers, mpcqp_dense.h)
//     (B) owners -> solution vector                    | barrier |  owners: relaxation, projection, dual step of their rows
//     (C) owners -> W = omega z - c y of their rows    | barrier |  next (A) reads the neighbours' W
// Slot v = k (nx+nu) + a of the compact order is owned by the LANE PAIR (2v, 2v+1) -- the pair that computes row v of the
// mat-vec, so the solution arrives in the owners' registers by itself -- and the pair splits the slot's work and registers:
//     state x_k[a]:  even lane  x, its dynamics row, row a of [Ad Bd]   |  odd lane  the slack eps, the state-box row, column a of Ad
//     input u_k[j]:  even lane  u, its box row (+ the first-step row)    |  odd lane  the Delta-u row, column j of Bd
// (one coefficient vector and at most two rows per lane; the two halves of a right-hand-side entry meet through one DPP swap).
// Same arithmetic as own_rhs / own_update, same results (tested against the block sweeps and against the oracle).
#pragma once

struct TinyRow { double z, ys, w, om, lo, hi; };              // one constraint row: z, c y / omega, W = omega (z - ys), omega, bounds
__device__ __forceinline__ void tiny_row_load(TinyRow &r, cgdouble *gz, cgdouble *gy, cgdouble *om, double cc, int idx, double lo, double hi) {
    r.om = om[idx]; r.z = gz[idx]; r.ys = cc * gy[idx] / r.om; r.w = r.om * (r.z - r.ys);
    r.lo = lo < -QP_INFTY ? -QP_INFTY : lo; r.hi = hi > QP_INFTY ? QP_INFTY : hi;
}
// relaxation, projection on [lo, hi], dual step (own_update's `row`); returns the dual increment in y units
__device__ __forceinline__ double tiny_row_step(TinyRow &r, double zt, double alpha, double beta, double cinv) {
    const double zr = alpha * zt + beta * r.z;
    const double zn = fmin(fmax(zr + r.ys, r.lo), r.hi);
    const double d = zr - zn;
    r.ys += d; r.z = zn; r.w = r.om * (zn - r.ys);
    return (r.om * cinv) * d;
}

// MAXB: length the coefficient vectors are zero-padded to (nx + nu <= MAXB <= 16; 8 for the reference's own small examples: half the LDS reads and FMAs of the two 16-long products)
template <int NB, int MAXB>
__device__ __forceinline__ void admm_tiny(const Lay &L, const HotPtrs &P, Smem &S, double *Xl, double *Zl, double *Yl, double alpha, int iters) {
    constexpr int VPAD = 16;                                  // padding of the exchange vectors (a lane's reads run up to MAXB entries past its stage)
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const int nx = L.nx, nu = L.nu, nb = L.nb;
    gdouble *gx = (gdouble *)(P.x + (size_t)b * L.n), *gz = (gdouble *)(P.z + (size_t)b * L.m), *gy = (gdouble *)(P.y + (size_t)b * L.m);
    cgdouble *om = (cgdouble *)(P.omega + (size_t)b * L.m), *sv = (cgdouble *)(P.s + (size_t)b * L.n), *qv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double cc = P.c[b], cinv = 1.0 / cc, beta = 1.0 - alpha;
    const double *hot = S.hot;
    TICK_RESET
    TICK_START
    // LDS exchange vectors (in the work area T; nothing else lives there during a round)
    double *Cv = S.T, *Xt = Cv + DenseFmt::CV, *Wd = Xt + DenseFmt::ROWS + VPAD, *Wu = Wd + DenseFmt::ROWS + VPAD;
    for (int i = tid; i < DenseFmt::SCRATCH; i += NT) S.T[i] = 0.0;
    // ---- who am I
    const int v = tid >> 1, k = v / nb, a = v - k * nb;
    const bool odd = tid & 1, live = v < L.NR;
    const bool is_x = live && a < nx, is_u = live && a >= nx && k < L.Nc;
    const int e = k * nx + a, cu = k * nu + (a - nx), jj = a - nx;
    const bool first_u = is_u && cu < nu;
    // ---- round-resident registers of this lane: one variable, up to two rows, one coefficient vector
    double pv = 0.0, svp = 0.0, cq = 0.0, kap = 0.0;
    TinyRow r1{0, 0, 0, 1, 0, 0}, r2{0, 0, 0, 1, 0, 0};
    double cvec[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) cvec[i] = 0.0;
    int pidx = 0, r1idx = 0;                                  // where the variable and the first row live in x and in z / y
    const double cef = cc * hot[L.oeps];
    if (is_x && !odd) {                                       // x_k[a], dynamics row e, row a of [Ad Bd]
        pidx = e; r1idx = e;
        pv = gx[e]; svp = sv[e]; cq = cc * qv[e];
        const double b0 = k == 0 ? -S.x0s[a] : 0.0;
        tiny_row_load(r1, gz, gy, om, cc, e, b0, b0);
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) cvec[i] = i < nx ? hot[L.oAd + a * nx + i] : (i < nb ? hot[L.oBd + a * nu + (i - nx)] : 0.0);
        }
    }
    if (is_x && odd) {                                        // eps_k[a], state-box row (soft: x + eps), column a of Ad
        pidx = L.oe + e; r1idx = L.rs + e;
        if (L.soft) { pv = gx[L.oe + e]; svp = sv[L.oe + e]; }
        tiny_row_load(r1, gz, gy, om, cc, L.rs + e, hot[L.oxmin + a], hot[L.oxmax + a]);
        kap = 1.0 / (cef + svp + r1.om);
        if (k < L.Np) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) if (i < nx) cvec[i] = hot[L.oAd + i * nx + a];
        }
    }
    if (is_u && !odd) {                                       // u_k[jj], input box row, first-step row u_0 - u_{-1}
        pidx = L.ou + cu; r1idx = L.ri + cu;
        pv = gx[L.ou + cu]; svp = sv[L.ou + cu]; cq = cc * qv[L.n_x + cu];
        tiny_row_load(r1, gz, gy, om, cc, L.ri + cu, hot[L.oumin + jj], hot[L.oumax + jj]);
        if (first_u) tiny_row_load(r2, gz, gy, om, cc, L.rdu + cu, S.du0[cu], S.du0[nu + cu]);
    }
    if (is_u && odd) {                                        // Delta-u row nu + cu (mpc.py:570), column jj of Bd
        r1idx = L.rdu + nu + cu;
        tiny_row_load(r1, gz, gy, om, cc, L.rdu + nu + cu, hot[L.oDumin + jj], hot[L.oDumax + jj]);
#pragma unroll
        for (int i = 0; i < MAXB; ++i) if (i < nx) cvec[i] = hot[L.oBd + i * nu + jj];
    }
    // what the coefficient vector multiplies: odd lanes the next stage's dynamics-row W (A'W), even lanes the previous stage's solution
    const double *nbr = odd ? Wd + min((k + 1) * nx, L.n_x) : Xt + max(k - 1, 0) * nb;
    // Nc < Np (mpc.py:513-517,540-543): the last input is held to the end of the horizon.  (i) The dynamics row of a stage behind it takes its
    // input part from slot Nc - 1.  (ii) The held input's A'W sums B' W_dyn over every later stage: the term of stage s > Nc is computed by the
    // otherwise idle odd lane of the input slot of stage s (same instruction stream as every other A'W term) and lands in the right-hand side
    // at ITS slot -- whose column of the stored inverse is a copy of the held input's column (factor_dense), so the mat-vec adds the terms up.
    const bool held = L.Nc < L.Np;
    const double *nbu = Xt + min(max(k - 1, 0), L.Nc - 1) * nb + nx;           // input part for the even x lanes
    const bool hterm = held && live && odd && a >= nx && k > L.Nc;
    if (hterm) {
#pragma unroll
        for (int i = 0; i < MAXB; ++i) if (i < nx) cvec[i] = hot[L.oBd + i * nu + jj];
        nbr = Wd + k * nx;
    }
    const int unext = (jj + 1 < nu) ? v + 1 : (k + 1) * nb + nx;      // slot of the next flattened input (cu + 1 < n_u)
    const bool has_unext = is_u && cu + 1 < L.n_u;
    const bool has_uprev = is_u && cu > 0;
    const int cvi = dense_cv_index(v);
    // (the inverse is requested BEHIND the owners' values: the memory counter retires in order, and 64 loads in front of them kept the first exchange waiting for
    //  the whole 128 KB; now it streams in under the two barriers and the first right-hand side)
    double Kreg[DenseFmt::JW];
    dense_load(P.F + (size_t)b * P.fsz, Kreg);
    __syncthreads();
    if (is_x && !odd) Wd[e] = r1.w;
    if (is_u && odd) Wu[cu] = r1.w;
    __syncthreads();
    auto dot = [&](bool split_u) {                            // cvec . nbr[0..MAXB): all reads issued together, then the FMAs
        double t[MAXB];
        if (split_u) {                                        // (Nc < Np, phase B: the input part of [x_prev | u_prev] may sit in another slot)
#pragma unroll
            for (int i = 0; i < MAXB; ++i) t[i] = (odd || i < nx) ? nbr[i] : nbu[i - nx];
        } else {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) t[i] = nbr[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int i = 0; i < MAXB; i += 2) { acc0 = fma(cvec[i], t[i], acc0); acc1 = fma(cvec[i + 1], t[i + 1], acc1); }
        return acc0 + acc1;
    };
    // Both lanes of a pair run ONE instruction stream per phase (a dot product, selects, one row step): roles differ in the
    // operands, not in the code -- a branch per role would execute both sides one after the other in every wave.
    const bool odd_x = odd && is_x, odd_u = odd && is_u, even_x = !odd && is_x, even_u = !odd && is_u, any = is_x || is_u;
    const bool slack = odd_x && L.soft;
    const double *unp = has_unext && odd ? Xt + unext : Cv + DenseFmt::CV - 1;      // next flattened input (or a zero that is never written)
    const double *upp = has_uprev && odd ? Wu + (cu - 1) : Cv + DenseFmt::CV - 1;  // Delta-u row of the previous flattened input
    const double s_uprev = has_uprev && odd ? 1.0 : 0.0, s_unext = has_unext && odd ? 1.0 : 0.0;
    const double okap = odd_x ? r1.om * kap : 0.0;
    TICK(7)
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;
        TICK_START
        // ---- (A) right-hand side  s x - c q + A'W  with the slack eliminated: the pair's two halves
        //   even x: s x - c q - W_dyn      odd x: A'W + (W_box - omega te)      even u: s u - c q + W_box (+ W_du0)      odd u: B'W - W_du + W_du(prev)
        const double atw = dot(false);
        const double wprev = *upp;
        const double te = slack ? (svp * pv + r1.w) * kap : 0.0;
        double part = odd ? atw : svp * pv - cq;
        part += odd_x ? r1.w - r1.om * te : (even_u ? r1.w + r2.w : -r1.w);
        part = fma(s_uprev, wprev, part);
        part = (any || hterm) ? part : 0.0;
        const double rhs = part + lane_swap1(part);
        if (!odd && live) Cv[cvi] = rhs;
        __syncthreads();
        TICK(0)
        // ---- the solve: row v of K^-1 times the right-hand side (both lanes of the pair get it)
        const double xt = dense_row<true>(Kreg, nullptr, Cv);
        if (!odd && live) Xt[v] = xt;
        __syncthreads();
        TICK(1)
        // ---- (B) relaxation, projection, dual step
        //   even x: x, dynamics row (zt = [Ad Bd] v_prev - xt)   odd x: eps, box row (zt = xt + et)   even u: u, box row (zt = ut), first-step row   odd u: Delta-u row
        const double gv = dot(held);
        const double un = *unp;
        const double et = slack ? te - okap * xt : 0.0;
        const double vt = odd ? et : xt;                      // the variable this lane updates moves towards vt
        const double vn = alpha * vt + beta * pv;
        const bool has_var = !odd ? any : slack;
        if (keep_delta && has_var) dxg[pidx] = vn - pv;
        pv = has_var ? vn : pv;
        const double zt = even_x ? gv - xt : (odd_x ? xt + et : (even_u ? xt : fma(s_unext, un, -xt)));
        const double d1 = tiny_row_step(r1, zt, alpha, beta, cinv);
        if (keep_delta && any) dyg[r1idx] = d1;
        if (first_u && !odd) { const double d2 = tiny_row_step(r2, xt, alpha, beta, cinv); if (keep_delta) dyg[L.rdu + cu] = d2; }
        if (even_x) Wd[e] = r1.w;
        if (odd_u) Wu[cu] = r1.w;
        __syncthreads();
        TICK(5)
    }
    TICK_START
    // ---- end of the round: the iterate back to memory (global: the next round / warm start; LDS copy: the residual evaluation)
    auto put_row = [&](const TinyRow &r, int idx) { const double y = r.ys * (r.om * cinv); gz[idx] = r.z; gy[idx] = y; Zl[idx] = r.z; Yl[idx] = y; };
    if (is_x || is_u) {
        if (!(odd && (is_u || !L.soft))) { gx[pidx] = pv; Xl[pidx] = pv; }
        put_row(r1, r1idx);
        if (first_u && !odd) put_row(r2, L.rdu + cu);
    }
    TICK(9)
    TICK_FLUSH
}

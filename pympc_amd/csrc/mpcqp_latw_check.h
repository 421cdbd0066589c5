// mpcqp_latw_check.h -- part of libmpcqp_hip (included by mpcqp.hip and mpcqp_w8.hip behind mpcqp_kernels.h).
// OSQP's termination test (update_info + check_termination: what check_body of mpcqp_phases.h evaluates on the iterate's LDS copy) in the OWNER
// layout of the latency round (mpcqp_latw.h: lane (I, B, J) of wave w owns slot a = 4B + I of stage s = 4w + J -- one group of four stages per
// wave), called by admm_latw between two rounds with the owned values BY VALUE:
//   * A x and A'y are the two MFMA groups of the iteration (G v of the previous stage, G' y_dyn of the next) applied to x and y, P x a 16-long dot
//     product per lane against the weight matrices' LDS copy; seven max-norms, the objective and the cheap halves of both infeasibility certificates
//     (|c dy| clipped by the infinite bounds and the support function of [l, u]; |dx| and q'dx) meet in ONE block reduction;
//   * LATW_SOLVED: the solve is finished here -- iterate for the next warm start, reported solution, mpcqp_info, statistics, first input and status
//     for the closed loop on the device: everything check_body's tail writes;
//   * LATW_CONTINUE: not converged and neither certificate can hold: the caller starts the next round at once, fragments and owner registers
//     where they are;
//   * LATW_GENERIC: anything else (a certificate that needs its operator product, a non-finite residual, the first launch of a two-launch
//     solve): the caller writes the iterate back and the generic check does what it always did.
// A function of its own on purpose: see the head of admm_latw.  Per check of one (12,4,30) instance ~ 4 k cycles against 15 k of the generic one
// plus 10 k of write-back and reload around it.
#pragma once

template <int NXT, int NUT, int NST>
__device__ __noinline__ int latw_check(double pv, double pv2, double ncq, double zA, double ysA, double omA, double zB, double ysB, double omB, double z0, double ys0, double om0,
                                       double loA, double hiA, double loB, double hiB, double cc, int iter, int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    constexpr int NB = 16, N = NST, NG = (N + 3) / 4, VS = LAT_VS(N), NTOP = BcrFmt::top_count(N);
    static_assert(NG <= NWAVES, "one group of four stages per wave");
    const RunKArgs &KA = run_kargs();
    const Lay &L = KA.L; const Ptrs &PP = KA.P; const RunArgs &R = KA.R;
    RunSmem rs = run_smem<true>(L, PP);
    Smem &S = rs.S;
    const int nx = NXT ? NXT : L.nx, nu = NUT ? NUT : L.nu, NR = L.N;
    const int b = inst_of(PP.perm), tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double cinv = 1.0 / cc;
    const double *hot = S.hot;
    double *Tc = S.T + NB, *Cc = Tc + VS;
    const double *TopL = rs.Y + L.m + ((smem_common_doubles(L) + L.n + 2 * L.m) & 1);
    // ---- the owner map, as in admm_latw
    const int a = 4 * ((lane >> 2) & 3) + (lane >> 4), J = lane & 3;
    const bool is_x = a < nx;
    const int jj = a - nx;
    const int s = wv < NG ? 4 * wv + J : 4 * NG, sl = s * NB + a;
    const bool ok = is_x ? s < NR : (a < nx + nu && s < NR - 1);
    const int e = s * nx + a, cu = s * nu + jj;
    const int pidx = is_x ? e : L.ou + cu, aidx = is_x ? e : L.ri + cu, bidx = is_x ? L.rs + e : L.rdu + nu + cu;
    const bool u0v = wv == 0 && J == 0 && !is_x && a < nx + nu;
    const int onext = (jj + 1 < nu) ? 1 : NB - nu + 1, oprev = (jj > 0) ? -1 : -(NB - nu + 1);
    // (the last iteration's increments: each lane reads back what it stored itself for the generic check -- requested first, consumed last)
    cgdouble *dxg = (cgdouble *)(PP.dx + (size_t)b * L.n), *dyg = (cgdouble *)(PP.dy + (size_t)b * L.m);
    const int ip = ok ? pidx : 0;
    const double ddx = dxg[ip], dde = dxg[(ok && is_x && L.soft) ? ip + L.oe : 0], ddA = dyg[ok ? aidx : 0], ddB = dyg[ok ? bidx : 0];
    const double dd0 = dyg[L.rdu + ((jj >= 0 && jj < nu) ? jj : 0)];
    // which of the owned rows' bounds are infinite on OSQP's scaled test (E hi > 1e26, E lo < -1e26: the primal infeasibility certificate clips the
    // dual increment by them) -- two bits per row: rA 0,1  rB 2,3  r0 4,5
    cgdouble *Eg = (cgdouble *)(PP.E + (size_t)b * L.m);
    const double eA = Eg[ok ? aidx : 0], eB = Eg[ok ? bidx : 0], e0 = Eg[L.rdu + ((jj >= 0 && jj < nu) ? jj : 0)], lim = QP_INFTY * MIN_SCALING;
    // x in Tc, the first rows' multipliers in Cc -- and, once G'y has been formed, the second rows' in Cc again (every slot 0 .. 4 NG - 1 has an
    // owner that writes it; slots -1 and 4 NG stay zero throughout; the iteration rewrites both vectors before it reads them)
    const double yA = ysA * (omA * cinv), yB = ysB * (omB * cinv), y0 = u0v ? ys0 * (om0 * cinv) : 0.0;
    Tc[sl] = ok ? pv : 0.0; Cc[sl] = ok ? yA : 0.0;
    __syncthreads();
    int infbits = !ok ? 0 : (eA * hiA > lim ? 1 : 0) | (eA * loA < -lim ? 2 : 0) | (eB * hiB > lim ? 4 : 0) | (eB * loB < -lim ? 8 : 0);
    if (u0v) infbits |= (e0 * S.du0[nu + jj] > lim ? 16 : 0) | (e0 * S.du0[jj] < -lim ? 32 : 0);
    const Ctx c{L, hot, hot + L.hot_sz};
    // 0 pri, 1 |Ax|, 2 |z|, 3 dua, 4 |Px|, 5 |A'y|, 6 |q|, 7 |c dy| clipped, 8 |dx|;  sums: objective, support function of [l, u] at c dy, q'dx
    double nrm[9], vsum[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) nrm[i] = 0.0;
    vsum[0] = vsum[1] = vsum[2] = 0.0;
    auto row = [&](double ax, double z) { nrm[0] = fmax(nrm[0], fabs(ax - z)); nrm[1] = fmax(nrm[1], fabs(ax)); nrm[2] = fmax(nrm[2], fabs(z)); };
    auto var = [&](double px, double aty, double qj, double xj, double dxj) {
        nrm[3] = fmax(nrm[3], fabs(px + qj + aty)); nrm[4] = fmax(nrm[4], fabs(px)); nrm[5] = fmax(nrm[5], fabs(aty)); nrm[6] = fmax(nrm[6], fabs(qj));
        vsum[0] += xj * (0.5 * px + qj);
        nrm[8] = fmax(nrm[8], fabs(dxj)); vsum[2] += qj * dxj;
    };
    auto cert = [&](double dy, double lo, double hi, int bits) {      // OSQP is_primal_infeasible: c dy projected on the polar of the recession cone of [l, u]
        double v = cc * dy;
        if (bits & 1) v = (bits & 2) ? 0.0 : fmin(v, 0.0);
        else if (bits & 2) v = fmax(v, 0.0);
        nrm[7] = fmax(nrm[7], fabs(v)); vsum[1] += hi * fmax(v, 0.0) + lo * fmin(v, 0.0);
    };
    double g = 0.0, h2 = 0.0, gt = 0.0, ht = 0.0;
    lat_mv(latw_top_frag(TopL, NTOP * NTOP, lane), Tc[sl - NB], g, h2);            // G v of the previous stage
    lat_mv(latw_top_frag(TopL, NTOP * NTOP + 1, lane), Cc[sl + NB], gt, ht);      // G' y_dyn of the next stage
    const double un = Tc[sl + onext];
    __syncthreads();
    Cc[sl] = ok ? yB : 0.0;
    __syncthreads();
    if (ok) {
        const double xj = pv, qj = -ncq * cinv, yprev = Cc[sl + oprev];
        const double *xs = Tc + s * NB;
        if (is_x) {
            row((g + h2) - xj, zA);
            row(L.soft ? xj + pv2 : xj, zB);
            const double *Q = (s < L.Np) ? c.Qx() : c.QxN();
            double px = 0.0;
#pragma unroll 4
            for (int l = 0; l < nx; ++l) px += Q[min(a, l) * nx + max(a, l)] * xs[l];
            var(px, (gt + ht) - yA + yB, qj, xj, ddx);
            if (L.soft) var(c.eps_feas() * pv2, yB, 0.0, pv2, dde);
        } else {
            row(xj, zA);
            row(un - xj, zB);
            const double iu = (s == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0, dk = (s == L.Nc - 1) ? 1.0 : 2.0;
            const double *Qu = c.Qu(), *QDu = c.QDu(), *us = xs + nx;
            double px = 0.0;
            for (int l = 0; l < nu; ++l) { const int lo = min(jj, l), hi = max(jj, l); px += input_weight(iu, Qu[lo * nu + hi], dk, QDu[lo * nu + hi]) * us[l]; }
            if (s + 1 < L.Nc) for (int l = 0; l < nu; ++l) px += -QDu[jj * nu + l] * us[NB + l];
            if (s > 0) for (int l = 0; l < nu; ++l) px += -QDu[l * nu + jj] * us[l - NB];
            double aty = (gt + ht) + yA - yB + yprev;
            if (u0v) { aty += y0; row(xj, z0); cert(dd0, S.du0[jj], S.du0[nu + jj], infbits >> 4); }
            var(px, aty, qj, xj, ddx);
        }
        cert(ddA, loA, hiA, infbits); cert(ddB, loB, hiB, infbits >> 2);
    }
    block_reduce<9, 3>(nrm, vsum, S.red);
    const double ea = KA.S.eps_abs, er = KA.S.eps_rel, epi = KA.S.eps_prim_inf, edi = KA.S.eps_dual_inf;
    const double pri_res = nrm[0], dua_res = nrm[3], obj_val = vsum[0];
    if (!(pri_res <= QP_INFTY) || !(dua_res <= QP_INFTY) || obj_val != obj_val) return LATW_GENERIC;      // (non-finite: the generic check reports it)
    const bool pc = pri_res < ea + er * fmax(nrm[2], nrm[1]), dc = dua_res < ea + er * fmax(fmax(nrm[6], nrm[5]), nrm[4]);
    if (pc && dc) {
        // ---- solved: what check_body's tail writes (the iterate for the next warm start, the reported solution, the record)
        gdouble *gx = (gdouble *)(PP.x + (size_t)b * L.n), *gz = (gdouble *)(PP.z + (size_t)b * L.m), *gy = (gdouble *)(PP.y + (size_t)b * L.m);
        gdouble *xo = (gdouble *)(PP.xo + (size_t)b * L.n), *yo = (gdouble *)(PP.yo + (size_t)b * L.m);
        if (ok) {
            gx[pidx] = pv; xo[pidx] = pv;
            if (is_x && L.soft) { gx[pidx + L.oe] = pv2; xo[pidx + L.oe] = pv2; }
            gz[aidx] = zA; gy[aidx] = yA; yo[aidx] = yA;
            gz[bidx] = zB; gy[bidx] = yB; yo[bidx] = yB;
        }
        if (u0v) { gz[L.rdu + jj] = z0; gy[L.rdu + jj] = y0; yo[L.rdu + jj] = y0; S.uo[jj] = pv; }
        if (tid == 0) {
            mpcqp_info inf;
            inf.status = MPCQP_SOLVED; inf.iter = iter; inf.rho_updates = S.iflag[1]; inf.reserved = (S.iflag[3] += 1);
            inf.obj_val = obj_val; inf.pri_res = pri_res; inf.dua_res = dua_res; inf.rho = PP.rho[b];
            S.iflag[4] = MPCQP_SOLVED;
            atomicAdd(&PP.stats[0], (unsigned long long)iter); atomicAdd(&PP.stats[1], (unsigned long long)inf.reserved);
            atomicAdd(&PP.stats[2], (unsigned long long)inf.rho_updates); atomicAdd(&PP.stats[3], 1ULL);
            PP.work[b] += (unsigned)iter;
            inf.reserved = 0;
            PP.info[b] = inf;
        }
        return LATW_SOLVED;
    }
    // not converged: may the next round start at once?  Only if neither certificate can hold -- decided from their cheap halves
    if (R.nsteps == 0 && R.part == 1) return LATW_GENERIC;                       // (the first launch of a two-launch solve hands over after ONE round)
    if (!pc && nrm[7] > epi && vsum[1] < -epi * nrm[7]) return LATW_GENERIC;     // primal infeasibility would need |A' dy|
    if (!dc && nrm[8] > edi && vsum[2] < -edi * nrm[8]) return LATW_GENERIC;     // dual infeasibility would need |P dx| and A dx
    if (tid == 0) S.iflag[3] += 1;
    return LATW_CONTINUE;
}

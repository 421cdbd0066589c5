// mpcqp_phases.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// Setup kernel and the phases of a solve: begin, check (OSQP termination / infeasibility / rho adaptation), ADMM iterations.
#pragma once

// rho vector -> metric.  Constraint types are decided on the SCALED bounds, as OSQP does.
__device__ __forceinline__ int row_type(double E, double lo, double hi) {
    double ls = E * lo, us = E * hi;
    if (ls < -QP_INFTY * MIN_SCALING && us > QP_INFTY * MIN_SCALING) return -1;
    if (us - ls < RHO_TOL) return 1;
    return 0;
}
__device__ __forceinline__ double row_rho(int type, double rho) { return type < 0 ? RHO_MIN : (type > 0 ? RHO_EQ_OVER_RHO_INEQ * rho : rho); }

// Shared prologue: stage the hot model prefix and the step data in LDS.
struct Smem {
    double *T, *Qv, *hot, *x0s, *um1s, *du0, *red, *tv, *xrs, *uo;
    int *iflag;
};
// Smem::iflag (ints): [0] a factorization's non-positive-pivot flag, [1] rho updates and [3] termination checks of the solve in progress (what
// mpcqp_info reports: kept here so that a check writes the record without reading it back first), [2] the latency round's LDS copy of the top
// inverse is valid (mpcqp_latw.h), [4] status of the solve that just ended.  Smem::uo: its first input (what output() returns), Smem::xrs: the
// constant reference (xref_rows == 1) -- both for the next step of the closed loop on the device, which would otherwise fetch them from memory.
__device__ __forceinline__ BorderPtrs border_ptrs(const Lay &L, const Ptrs &P, const Smem &S) {
    BorderPtrs bp; bp.red = S.red;
    const size_t npb = (size_t)L.nu * L.N * L.NB;
    const int b = inst_of(P.perm);
    bp.gws = L.NB == 128 ? P.bws + (size_t)b * HugeFmt::GWS : nullptr;
    bp.Bb = L.border ? P.Bb + b * npb : nullptr;
    bp.Zb = L.border ? P.Zb + b * npb : nullptr;
    bp.Sig = L.border ? P.Sig + (size_t)b * L.nu * L.nu : nullptr;
    return bp;
}

__device__ __forceinline__ double *carve(double *&p, int n) { double *r = p; p += n; return r; }
// LDS layout of a workgroup: [ hot | x0 | um1 | du0 | red | tv | iflag | T (tsz) ] and, behind it, the LDS-resident iterate
// [ X | Z | Y ] of small problems.  T comes last so that the factorization, which runs while the iterate copy is dead
// (before a round loads it, after a check has read it), can let its workspace run on into that area.
template <class PT>
__device__ __forceinline__ void smem_common(const Lay &L, const PT &P, double *&p, Smem &S) {
    S.Qv = (double *)P.qv + (size_t)inst_of(P.perm) * (L.n_x + L.n_u);
    S.hot = carve(p, L.hot_lds);
    S.x0s = carve(p, L.nx);
    S.um1s = carve(p, L.nu);
    S.du0 = carve(p, 2 * L.nu);
    S.red = carve(p, 16 * L.nw);   // block_reduce: up to 12 values per wave
    S.tv = carve(p, 128);          // the bordered solve's ubar (nu <= 127 doubles); outside it, the held input's A'W sums (nu)
    S.xrs = carve(p, L.nx);
    S.uo = carve(p, L.nu);
    S.iflag = (int *)carve(p, 4);
    S.T = carve(p, L.tsz);
}
__host__ __device__ inline int smem_common_doubles(const Lay &L) { return L.tsz + L.hot_lds + 2 * L.nx + 4 * L.nu + 16 * L.nw + 128 + 4; }

__device__ __forceinline__ void load_common(const Lay &L, const double *model, const double *step, Smem &S) {
    for (int i = threadIdx.x; i < L.hot_lds; i += NT) S.hot[i] = model[i];
    for (int i = threadIdx.x; i < L.nx; i += NT) { S.x0s[i] = step[i]; S.xrs[i] = step[L.nx + L.nu + i]; }      // (xrs: row 0 of the reference; used where it is the only row)
    for (int i = threadIdx.x; i < L.nu; i += NT) {
        const double um1 = step[L.nx + i];
        S.um1s[i] = um1;
        // first Delta-u rows: Dumin/Dumax + u_{-1} (mpc.py:407-408), or verbatim what mpcqp_update_vectors was given
        S.du0[i] = L.raw ? step[L.odu0 + i] : model[L.oDumin + i] + um1;
        S.du0[L.nu + i] = L.raw ? step[L.odu0 + L.nu + i] : model[L.oDumax + i] + um1;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// setup kernel: Ruiz equilibration (OSQP, 10 passes), rho vector, metric, first factorization.
// ------------------------------------------------------------------------------------------------
// Two launches.  k_setup: everything but the factorization -- ten passes of dependent global round trips (latency, hidden by co-resident workgroups: 52 registers,
// the common block WITHOUT the work area T, a few KB of LDS); k_setup_factor<NB>: the first factorization with its own registers and LDS.  (One kernel until round 6:
// 219 registers and, for the cyclic reduction with a dense top, 142 KB of LDS held the equilibration passes to two / one workgroup per compute unit.)
// lds_de: the scaling vectors D, E live in LDS during the passes (where the work area T would start; the launch provides n + m doubles there) -- every term of
// every row norm reads one of them, through a visitor the compiler cannot hoist the loads out of: a memory round trip per term otherwise.
__global__ __launch_bounds__(NT) void k_setup(Lay L, Ptrs P, mpcqp_settings S_, int lds_de) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    if (!L.raw) build_q(c, step, S.Qv, S.um1s, L.xref_rows == 1 ? S.xrs : nullptr);      // (raw vectors: q was uploaded by mpcqp_update_vectors)
    double *Dg = P.D + (size_t)b * L.n, *Eg = P.E + (size_t)b * L.m, *Dt = P.Dt + (size_t)b * L.n, *Et = P.Et + (size_t)b * L.m;
    double *D = lds_de ? S.T : Dg, *E = lds_de ? S.T + L.n : Eg;
    for (int j = tid; j < L.n; j += NT) D[j] = 1.0;
    for (int r = tid; r < L.m; r += NT) E[r] = 1.0;
    double cc = 1.0;
    __syncthreads();
    for (int it = 0; it < S_.scaling; ++it) {
        for (int j = tid; j < L.n; j += NT) {
            double pn = 0.0, an = 0.0;
            P_row(c, j, [&](double co, int idx) { pn = fmax(pn, fabs(co) * D[idx]); });
            AT_row(c, j, [&](double co, int row) { an = fmax(an, fabs(co) * E[row]); });
            pn *= cc * D[j]; an *= D[j];
            Dt[j] = 1.0 / sqrt(limit_scaling(fmax(pn, an)));
        }
        for (int r = tid; r < L.m; r += NT) {
            double en = 0.0;
            A_row(c, r, [&](double co, int idx) { en = fmax(en, fabs(co) * D[idx]); });
            Et[r] = 1.0 / sqrt(limit_scaling(en * E[r]));
        }
        __syncthreads();
        for (int j = tid; j < L.n; j += NT) D[j] *= Dt[j];
        for (int r = tid; r < L.m; r += NT) E[r] *= Et[r];
        __syncthreads();
        double vmax[1] = {0.0}, vsum[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double pn = 0.0;
            P_row(c, j, [&](double co, int idx) { pn = fmax(pn, fabs(co) * D[idx]); });
            vsum[0] += cc * D[j] * pn;
            double qj = (j < L.oe) ? S.Qv[j] : 0.0;
            vmax[0] = fmax(vmax[0], fabs(cc * D[j] * qj));
        }
        block_reduce<1, 1>(vmax, vsum, S.red);
        double ct = vsum[0] / (double)L.n;
        double qn = limit_scaling(vmax[0]);
        ct = limit_scaling(fmax(ct, qn));
        cc *= 1.0 / ct;
    }
    if (lds_de) {
        for (int j = tid; j < L.n; j += NT) Dg[j] = D[j];
        for (int r = tid; r < L.m; r += NT) Eg[r] = E[r];
    }
    // rho vector / metric
    double rho = S_.rho;
    double *om = P.omega + (size_t)b * L.m, *sv = P.s + (size_t)b * L.n;
    int *ct = P.ctype + (size_t)b * L.m;
    for (int r = tid; r < L.m; r += NT) {
        double lo, hi; row_bounds(c, S.x0s, S.du0, r, lo, hi);
        int t = row_type(E[r], lo, hi);
        ct[r] = t;
        om[r] = row_rho(t, rho) * E[r] * E[r];
    }
    for (int j = tid; j < L.n; j += NT) sv[j] = S_.sigma / (D[j] * D[j]);
    if (tid == 0) { P.c[b] = cc; P.rho[b] = rho; }
    __syncthreads();
    // cold start
    for (int j = tid; j < L.n; j += NT) { P.x[(size_t)b * L.n + j] = 0.0; P.xo[(size_t)b * L.n + j] = 0.0; }
    for (int r = tid; r < L.m; r += NT) { P.z[(size_t)b * L.m + r] = 0.0; P.y[(size_t)b * L.m + r] = 0.0; P.yo[(size_t)b * L.m + r] = 0.0; }
    if (tid == 0) {
        mpcqp_info inf; inf.status = MPCQP_UNSOLVED; inf.iter = 0; inf.rho_updates = 0; inf.reserved = 0;      // (k_setup_factor: MPCQP_NON_CVX on a non-positive pivot)
        inf.obj_val = 0; inf.pri_res = 0; inf.dua_res = 0; inf.rho = rho;
        P.info[b] = inf;
    }
}

// the second launch of setup: the first factorization, from the metric k_setup left in memory
template <int NB>
__global__ __launch_bounds__(NT) void k_setup_factor(Lay L, Ptrs P) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm);
    const double *model = P.model + (size_t)b * L.model_sz;
    load_common(L, model, P.step + (size_t)b * L.step_sz, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    const double *om = P.omega + (size_t)b * L.m, *sv = P.s + (size_t)b * L.n;
    const double cc = P.c[b];
    const int bad = NB == 16 && L.grp > 1 ? factor_grouped(c, om, sv, cc, P.F + (size_t)b * P.fsz, S.T, S.iflag, border_ptrs(L, P, S))
                  : NB == 16 && L.dense ? factor_dense(c, om, sv, cc, P.F + (size_t)b * P.fsz, S.T, S.iflag)
                            : factor_all<NB>(c, om, sv, cc, P.F + (size_t)b * P.fsz, S.T, S.iflag, border_ptrs(L, P, S));
    if (bad && threadIdx.x == 0) P.info[b].status = MPCQP_NON_CVX;
}

// ... of the cyclic reduction (a kernel of its own: 130 registers and 47 KB of LDS at (12,4,30) -- three workgroups per compute unit)
__global__ __launch_bounds__(NT) void k_setup_factor_bcr(Lay L, Ptrs P) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm);
    const double *model = P.model + (size_t)b * L.model_sz;
    load_common(L, model, P.step + (size_t)b * L.step_sz, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    const int bad = factor_bcr(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, P.bws + (size_t)b * L.bcr * BcrFmt::WSTAGE, S.T, S.iflag);
    if (bad && threadIdx.x == 0) P.info[b].status = MPCQP_NON_CVX;
}

// ------------------------------------------------------------------------------------------------
// A solve has three phases (bodies below; k_mpc_run strings them together per instance):
//   begin  once per solve : q refresh from (x0, u_{-1}, xref), constraint types, per-solve bookkeeping
//   admm   per round      : check_termination ADMM iterations -- the hot loop, nothing else in it
//   check  per round      : residuals, termination, infeasibility certificates, rho adaptation + refactor
// ------------------------------------------------------------------------------------------------
enum { COLD_CHECK = 1, COLD_RHO = 2, COLD_FINAL = 4, COLD_PLAIN = 8 };

// Refactorization from inside a solve: a non-inlined function with a register allocation of its own (defined with the
// other phases of k_mpc_run below), so that this rare, register-hungry path does not push the residual evaluation
// and the per-solve prologue into scratch spills.
template <int NB, int OCC> __device__ void run_factor_phase(int *frame_pin);

template <int NB, int OCC>
__device__ __forceinline__ void begin_body(const Lay &L, const Ptrs &P, const mpcqp_settings &S_, Smem &S, int plain, int warm_x) {
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    Ctx c{L, S.hot, L.hot_lds > L.hot_sz ? S.hot + L.hot_sz : model + L.hot_sz};      // (the weight matrices: the LDS copy where there is one)
    if (!L.raw) build_q(c, step, S.Qv, S.um1s, L.xref_rows == 1 ? S.xrs : nullptr);
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    if (!(S_.warm_start || plain)) {
        for (int j = tid; j < L.n; j += NT) gx[j] = 0.0;
        for (int r = tid; r < L.m; r += NT) { gz[r] = 0.0; gy[r] = 0.0; }
    } else if (warm_x) {                                  // mpcqp_warm_start replaced x: z = A x, as osqp_warm_start does
        for (int r = tid; r < L.m; r += NT) { double ax = 0.0; A_row(c, r, [&](double co, int idx) { ax += co * gx[idx]; }); gz[r] = ax; }
    }
    // constraint types (bounds may have changed since the last factorization)
    double *om = P.omega + (size_t)b * L.m;
    const double *sv = P.s + (size_t)b * L.n, *E = P.E + (size_t)b * L.m;
    int *ctp = P.ctype + (size_t)b * L.m;
    const double rho = P.rho[b];
    int changed = 0;
    for (int r = tid; r < L.m; r += NT) {
        double lo, hi; row_bounds(c, S.x0s, S.du0, r, lo, hi);
        int t = row_type(E[r], lo, hi);
        if (t != ctp[r]) { changed = 1; ctp[r] = t; om[r] = row_rho(t, rho) * E[r] * E[r]; }
    }
    changed = __syncthreads_or(changed);
    if (changed) { int pin = 0; asm volatile("" : "+v"(pin)); run_factor_phase<NB, OCC>(&pin); asm volatile("" :: "v"(pin)); }
    if (tid == 0) {
        mpcqp_info inf; inf.status = MPCQP_UNSOLVED; inf.iter = 0; inf.rho_updates = 0; inf.reserved = 0;
        inf.obj_val = 0.0; inf.pri_res = 0.0; inf.dua_res = 0.0; inf.rho = rho;
        P.info[b] = inf;
        S.iflag[1] = 0; S.iflag[3] = 0;
    }
}

typedef __attribute__((address_space(3))) const double ldsd;      // an LDS view of a generic pointer that is known to point into LDS (its low 32 bits)
__device__ __forceinline__ ldsd *as_lds(const double *p) { return (ldsd *)(size_t)(unsigned)(unsigned long long)p; }

// Returns 1 (to every thread) if the instance has terminated.
// Residual norms and objective of the termination check for an LDS-resident iterate, with the owner map of the parallel phases below:
// thread t owns the state elements e = t, t + NT (the variable x_e, its slack, the dynamics row e and the state-box row) and the input
// element cu = t (the variable, its box row, its Delta-u row and, for cu < nu, its first-step row).  Every pass handles ONE kind of item --
// no tree over four row types and three variable kinds per visitor call -- and the thread's D, E, q values are fetched together up front
// instead of one global round trip per item.  Terms are summed in the order the row visitors (mpcqp_qp.h) enumerate them.
// nrm / vsum as in check_body.  Needs n_x <= 2 NT and n_u <= NT (what the LDS-resident mode is chosen by).
// NXT / NUT > 0: compile-time nx / nu (the one-workgroup-per-CU latency kernels, which have the registers for the unrolled sums: the same
// copy inside the four-per-CU kernels' 128 registers measured slower in round 3); 0: from the layout.
// scaled: also the scaled norms 7 .. 10, which only the rho estimate reads (every adaptive_rho_interval-th iteration: one check in four).
template <int NXT = 0, int NUT = 0>
__device__ __forceinline__ void check_norms_own(const Ctx &c, const double *Xg, const double *Zg, const double *Yg, const double *D, const double *E,
                                                const double *Qv, double cc, double *nrm, double *vsum, const bool scaled) {
    const Lay &L = c.L;
    constexpr int UNR = NXT ? 16 : 4;                  // (compile-time dimensions: the nx-long sums fully unrolled)
    const int tid = threadIdx.x, nx = NXT ? NXT : L.nx, nu = NUT ? NUT : L.nu;
    // Everything this pass reads a neighbour's value from sits in LDS -- the iterate's copy, the hot prefix, the weight matrices (staged with it,
    // or copied behind W by check_body: mpcqp_create sizes the work area so that they fit) -- but reaches this function through pointers the
    // compiler cannot place (selects of LDS and global bases): FLAT loads, 120 of them per thread, each waiting for every outstanding global
    // load as well.  Explicit LDS views: plain ds_read, and the D / E / q requests above stay in flight behind them.
    ldsd *X = as_lds(Xg), *Z = as_lds(Zg), *Y = as_lds(Yg), *Ad = as_lds(c.Ad()), *Bd = as_lds(c.Bd());
    auto row = [&](double ax, double z, double e) {
        const double d = ax - z;
        nrm[0] = fmax(nrm[0], fabs(d)); nrm[1] = fmax(nrm[1], fabs(ax)); nrm[2] = fmax(nrm[2], fabs(z));
        if (scaled) { nrm[7] = fmax(nrm[7], fabs(e * d)); nrm[8] = fmax(nrm[8], fmax(fabs(e * ax), fabs(e * z))); }
    };
    auto var = [&](double px, double aty, double qj, double xj, double cd) {
        const double d = px + qj + aty;
        nrm[3] = fmax(nrm[3], fabs(d)); nrm[4] = fmax(nrm[4], fabs(px)); nrm[5] = fmax(nrm[5], fabs(aty)); nrm[6] = fmax(nrm[6], fabs(qj));
        if (scaled) {
            nrm[9] = fmax(nrm[9], fabs(cd * d));
            nrm[10] = fmax(nrm[10], fmax(fabs(cd * qj), fmax(fabs(cd * aty), fabs(cd * px))));
        }
        vsum[0] += xj * (0.5 * px + qj);
    };
    double eDyn[2], eBox[2], dXe[2], dEe[2], qXe[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        const bool v = e < L.n_x;
        eDyn[j] = v ? E[e] : 0.0; eBox[j] = v ? E[L.rs + e] : 0.0; dXe[j] = v ? D[e] : 0.0; dEe[j] = (v && L.soft) ? D[L.oe + e] : 0.0; qXe[j] = v ? Qv[e] : 0.0;
    }
    const int cu = tid;
    const bool vu = cu < L.n_u;
    const double eIn = vu ? E[L.ri + cu] : 0.0, eDu = vu ? E[L.rdu + nu + cu] : 0.0, eD0 = cu < nu ? E[L.rdu + cu] : 0.0;
    const double dU = vu ? D[L.ou + cu] : 0.0, qU = vu ? Qv[L.n_x + cu] : 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (e < L.n_x) {
            const int k = NXT ? e / (NXT ? NXT : 1) : idiv(e, L.rnx), a = e - k * nx;
            const double xe = X[e], ee = L.soft ? X[L.oe + e] : 0.0;
            double ax = -xe;                                           // dynamics row e  (mpc.py:537-552)
            if (k > 0) {
                ldsd *xp = X + (k - 1) * nx, *up = X + L.ou + min(k - 1, L.Nc - 1) * nu;
#pragma unroll UNR
                for (int i = 0; i < nx; ++i) ax += Ad[a * nx + i] * xp[i];
#pragma unroll UNR
                for (int i = 0; i < nu; ++i) ax += Bd[a * nu + i] * up[i];
            }
            row(ax, Z[e], eDyn[j]);
            row(L.soft ? xe + ee : xe, Z[L.rs + e], eBox[j]);          // state-box row: x_k (+ eps_k)
            ldsd *Q = as_lds((k < L.Np) ? c.Qx() : c.QxN());
            ldsd *xk = X + k * nx;
            double px = 0.0, aty = -Y[e];
#pragma unroll UNR
            for (int l = 0; l < nx; ++l) px += Q[min(a, l) * nx + max(a, l)] * xk[l];
            if (k < L.Np) {
                ldsd *y1 = Y + (k + 1) * nx;
#pragma unroll UNR
                for (int r = 0; r < nx; ++r) aty += Ad[r * nx + a] * y1[r];
            }
            aty += Y[L.rs + e];
            var(px, aty, qXe[j], xe, cc * dXe[j]);
            if (L.soft) { double pe = 0.0, ae = 0.0; pe += c.eps_feas() * ee; ae += Y[L.rs + e]; var(pe, ae, 0.0, ee, cc * dEe[j]); }
        }
    }
    if (vu) {
        const int k = NUT ? cu / (NUT ? NUT : 1) : idiv(cu, L.rnu), jj = cu - k * nu;
        const double ut = X[L.ou + cu];
        row(ut, Z[L.ri + cu], eIn);                                    // input box
        double ax = -ut;                                               // Delta-u row nu + cu (mpc.py:570)
        if (cu + 1 < L.n_u) ax += X[L.ou + cu + 1];
        row(ax, Z[L.rdu + nu + cu], eDu);
        if (cu < nu) row(ut, Z[L.rdu + cu], eD0);                      // first step: u_0 (- u_{-1} in the bounds)
        const double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0, dk = (k == L.Nc - 1) ? 1.0 : 2.0;
        ldsd *Qu = as_lds(c.Qu()), *QDu = as_lds(c.QDu()), *uk = X + L.ou + k * nu;
        double px = 0.0;
        for (int l = 0; l < nu; ++l) {
            const int lo = min(jj, l), hi = max(jj, l);
            px += input_weight(iu, Qu[lo * nu + hi], dk, QDu[lo * nu + hi]) * uk[l];
        }
        if (k + 1 < L.Nc) for (int l = 0; l < nu; ++l) px += -QDu[jj * nu + l] * uk[nu + l];
        if (k > 0) for (int l = 0; l < nu; ++l) px += -QDu[l * nu + jj] * uk[l - nu];
        double aty = 0.0;
        const int s_end = (k == L.Nc - 1) ? L.Np : k + 1;            // the last input is held to the end of the horizon
        for (int s = k + 1; s <= s_end; ++s) {
            ldsd *y1 = Y + s * nx;
#pragma unroll UNR
            for (int r = 0; r < nx; ++r) aty += Bd[r * nu + jj] * y1[r];
        }
        aty += Y[L.ri + cu];
        if (k == 0) aty += Y[L.rdu + jj];
        aty += -Y[L.rdu + nu + cu];
        if (cu > 0) aty += Y[L.rdu + nu + cu - 1];
        var(px, aty, qU, ut, cc * dU);
    }
}

// The same norms for an iterate that lives in GLOBAL memory (problems too large for the LDS-resident round: cfg-5, long horizons, wide
// stages).  The row visitors used here before walk one row / variable at a time with a global load per term: 183 us of a 286 us check at
// cfg-5 while all 512 instances stream (an ADMM iteration: 111 us).  Instead the (x, u) part of x and all of y -- everything a NEIGHBOUR
// reads -- are staged once into the work vector T, which is idle during a check ([ Y (m) | X (n_x + n_u) ] fits: tsz >= m + N NB), with
// coalesced, independent loads; then the owner map of check_norms_own, as many passes as it takes (state element e = t + NT j, input
// element cu = t + NT j), two items per thread in flight with all their global operands (z, the slack, D, E, q) requested up front.
// Terms are summed in the order the row visitors enumerate them.  wts: the weight matrices (LDS copy or global).
__device__ __forceinline__ void check_norms_gown(const Ctx &c, const double *gX, const double *gZ, const double *gY, const double *D, const double *E,
                                                 const double *Qv, double cc, double *T, double *hsy, double *nrm, double *vsum, const bool scaled) {
    constexpr int GU = 2;
    const Lay &L = c.L;
    const int tid = threadIdx.x, nx = L.nx, nu = L.nu;
    ldsd *Ad = as_lds(c.Ad()), *Bd = as_lds(c.Bd());                   // (explicit LDS views, as in check_norms_own: plain ds_read instead of FLAT loads)
    cgdouble *Xg = (cgdouble *)gX, *Zg = (cgdouble *)gZ, *Yg = (cgdouble *)gY, *Dg = (cgdouble *)D, *Eg = (cgdouble *)E, *Qg = (cgdouble *)Qv;
    for (int i = tid; i < L.m; i += NT) T[i] = Yg[i];
    for (int i = tid; i < L.n_x + L.n_u; i += NT) T[L.m + i] = Xg[i];
    ldsd *Y = as_lds(T), *X = as_lds(T + L.m);                          // X: the (x, u) part only; a slack is read by its owner alone
    __syncthreads();
    // Nc < Np: the held input's part of A'y is a sum over the Np - Nc + 1 stages it acts on.  Its owner used to walk them alone (76 stages of
    // dependent LDS reads at the reference's Kalman notebook: 4 of a check's 20 us); the last wave forms it beside the state items -- lane l the
    // stages Nc + l, Nc + l + 64, ..., then a fixed-order butterfly -- and leaves the nu sums in hsy (LDS, nu doubles).
    const bool held = L.Nc < L.Np;
    if (held && tid >= NT - 64) {
        const int lane = tid - (NT - 64);
        for (int jj = 0; jj < nu; ++jj) {
            double a = 0.0;
            for (int s = L.Nc + lane; s <= L.Np; s += 64) {
                ldsd *y1 = Y + s * nx;
                double t = 0.0;
#pragma unroll 4
                for (int r = 0; r < nx; ++r) t += Bd[r * nu + jj] * y1[r];
                a += t;
            }
            a = wave_reduce<false>(a);
            if (lane == 0) hsy[jj] = a;
        }
    }
    auto row = [&](double ax, double z, double e) {
        const double d = ax - z;
        nrm[0] = fmax(nrm[0], fabs(d)); nrm[1] = fmax(nrm[1], fabs(ax)); nrm[2] = fmax(nrm[2], fabs(z));
        if (scaled) { nrm[7] = fmax(nrm[7], fabs(e * d)); nrm[8] = fmax(nrm[8], fmax(fabs(e * ax), fabs(e * z))); }
    };
    auto var = [&](double px, double aty, double qj, double xj, double cd) {
        const double d = px + qj + aty;
        nrm[3] = fmax(nrm[3], fabs(d)); nrm[4] = fmax(nrm[4], fabs(px)); nrm[5] = fmax(nrm[5], fabs(aty)); nrm[6] = fmax(nrm[6], fabs(qj));
        if (scaled) {
            nrm[9] = fmax(nrm[9], fabs(cd * d));
            nrm[10] = fmax(nrm[10], fmax(fabs(cd * qj), fmax(fabs(cd * aty), fabs(cd * px))));
        }
        vsum[0] += xj * (0.5 * px + qj);
    };
    for (int e0 = tid; e0 < L.n_x; e0 += GU * NT) {
        double eDyn[GU], eBox[GU], dXe[GU], dEe[GU], qXe[GU], zD[GU], zB[GU], ee[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = min(e0 + u * NT, L.n_x - 1);
            eDyn[u] = Eg[e]; eBox[u] = Eg[L.rs + e]; dXe[u] = Dg[e]; qXe[u] = Qg[e]; zD[u] = Zg[e]; zB[u] = Zg[L.rs + e];
            dEe[u] = 0.0; ee[u] = 0.0;
            if (L.soft) { dEe[u] = Dg[L.oe + e]; ee[u] = Xg[L.oe + e]; }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = e0 + u * NT;
            if (e < L.n_x) {
                const int k = idiv(e, L.rnx), a = e - k * nx;
                const double xe = X[e];
                double ax = -xe;                                           // dynamics row e  (mpc.py:537-552)
                if (k > 0) {
                    ldsd *xp = X + (k - 1) * nx, *up = X + L.ou + min(k - 1, L.Nc - 1) * nu;
#pragma unroll 4
                    for (int i = 0; i < nx; ++i) ax += Ad[a * nx + i] * xp[i];
#pragma unroll 4
                    for (int i = 0; i < nu; ++i) ax += Bd[a * nu + i] * up[i];
                }
                row(ax, zD[u], eDyn[u]);
                row(L.soft ? xe + ee[u] : xe, zB[u], eBox[u]);             // state-box row: x_k (+ eps_k)
                const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
                ldsd *xk = X + k * nx;
                double px = 0.0, aty = -Y[e];
#pragma unroll 4
                for (int l = 0; l < nx; ++l) px += Q[min(a, l) * nx + max(a, l)] * xk[l];
                if (k < L.Np) {
                    ldsd *y1 = Y + (k + 1) * nx;
#pragma unroll 4
                    for (int r = 0; r < nx; ++r) aty += Ad[r * nx + a] * y1[r];
                }
                aty += Y[L.rs + e];
                var(px, aty, qXe[u], xe, cc * dXe[u]);
                if (L.soft) { double pe = 0.0, ae = 0.0; pe += c.eps_feas() * ee[u]; ae += Y[L.rs + e]; var(pe, ae, 0.0, ee[u], cc * dEe[u]); }
            }
        }
    }
    if (held) __syncthreads();                                           // (the held input's sums)
    for (int c0 = NT - 1 - tid; c0 < L.n_u; c0 += GU * NT) {           // (input items from the last thread downwards, as in gown_rhs)
        double eIn[GU], eDu[GU], eD0[GU], dU[GU], qU[GU], zI[GU], zU[GU], z0[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int cu = min(c0 + u * NT, L.n_u - 1), r0 = L.rdu + min(cu, nu - 1);
            eIn[u] = Eg[L.ri + cu]; eDu[u] = Eg[L.rdu + nu + cu]; eD0[u] = Eg[r0]; dU[u] = Dg[L.ou + cu]; qU[u] = Qg[L.n_x + cu];
            zI[u] = Zg[L.ri + cu]; zU[u] = Zg[L.rdu + nu + cu]; z0[u] = Zg[r0];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int cu = c0 + u * NT;
            if (cu < L.n_u) {
                const int k = idiv(cu, L.rnu), jj = cu - k * nu;
                const double ut = X[L.ou + cu];
                row(ut, zI[u], eIn[u]);                                    // input box
                double ax = -ut;                                           // Delta-u row nu + cu (mpc.py:570)
                if (cu + 1 < L.n_u) ax += X[L.ou + cu + 1];
                row(ax, zU[u], eDu[u]);
                if (cu < nu) row(ut, z0[u], eD0[u]);                       // first step: u_0 (- u_{-1} in the bounds)
                const double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0, dk = (k == L.Nc - 1) ? 1.0 : 2.0;
                const double *Qu = c.Qu(), *QDu = c.QDu(); ldsd *uk = X + L.ou + k * nu;
                double px = 0.0;
                for (int l = 0; l < nu; ++l) {
                    const int lo = min(jj, l), hi = max(jj, l);
                    px += input_weight(iu, Qu[lo * nu + hi], dk, QDu[lo * nu + hi]) * uk[l];
                }
                if (k + 1 < L.Nc) for (int l = 0; l < nu; ++l) px += -QDu[jj * nu + l] * uk[nu + l];
                if (k > 0) for (int l = 0; l < nu; ++l) px += -QDu[l * nu + jj] * uk[l - nu];
                double aty = 0.0;
                if (held && k == L.Nc - 1) aty = hsy[jj];                  // the last input is held to the end of the horizon
                else {
                    ldsd *y1 = Y + (k + 1) * nx;
#pragma unroll 4
                    for (int r = 0; r < nx; ++r) aty += Bd[r * nu + jj] * y1[r];
                }
                aty += Y[L.ri + cu];
                if (k == 0) aty += Y[L.rdu + jj];
                aty += -Y[L.rdu + nu + cu];
                if (cu > 0) aty += Y[L.rdu + nu + cu - 1];
                var(px, aty, qU[u], ut, cc * dU[u]);
            }
        }
    }
    __syncthreads();                                                       // (T is reused by the certificates and by the caller)
}

template <int NB, int OCC>
__device__ __forceinline__ int check_body(const Lay &L, const Ptrs &P, const mpcqp_settings &S_, Smem &S, int iter, int mode,
                                          const double *Xl, const double *Zl, const double *Yl) {
    const int b = inst_of(P.perm), tid = threadIdx.x;
    TICK_RESET
    TICK_START
    const double *model = P.model + (size_t)b * L.model_sz;
    // the weight matrices (read entry by entry) go to the idle part of the work vector if they fit: behind W for the LDS-resident
    // iterate, behind the staged [ Y | X ] of check_norms_gown otherwise
    const int nweights = L.model_sz - L.hot_sz;
    const double *wts = model + L.hot_sz;
    const int wq_off = L.m + (Xl ? 0 : L.n_x + L.n_u);
    if (L.hot_lds > L.hot_sz) wts = S.hot + L.hot_sz;                     // (staged with the hot prefix when the kernel started: Lay::hot_lds)
    else if (nweights <= L.tsz - wq_off) {
        double *wq = S.T + wq_off;
        for (int i = tid; i < nweights; i += NT) wq[i] = model[L.hot_sz + i];
        __syncthreads();
        wts = wq;
    }
    Ctx c{L, S.hot, wts};
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    // the iterate: the LDS copy the last ADMM round left behind (small problems), else global memory
    const double *X = Xl ? Xl : gx, *Z = Zl ? Zl : gz, *Y = Yl ? Yl : gy;
    double *om = P.omega + (size_t)b * L.m;
    const double *sv = P.s + (size_t)b * L.n;
    const double *D = P.D + (size_t)b * L.n, *E = P.E + (size_t)b * L.m;
    const double *dxg = P.dx + (size_t)b * L.n, *dyg = P.dy + (size_t)b * L.m;
    const int *ctp = P.ctype + (size_t)b * L.m;
    const double cc = P.c[b];
    double rho = P.rho[b];
    int status = MPCQP_UNSOLVED;
    double obj_val, pri_res, dua_res;

    // ---- OSQP update_info: objective, unscaled residuals, and the scaled norms the rho estimate needs
    // vmax: 0 pri, 1 |Ax|, 2 |z|, 3 dua, 4 |Px|, 5 |A'y|, 6 |q|; scaled: 7 pri, 8 max(|EAx|,|Ez|), 9 dua, 10 max(|cD(..)|)
    double nrm[11], vsum[1] = {0.0};
    const bool scaled = (mode & COLD_RHO) != 0;                           // (the scaled norms 7 .. 10 feed the rho estimate alone)
#pragma unroll
    for (int i = 0; i < 11; ++i) nrm[i] = 0.0;
    TICK(10)
    if (Xl) {                                                             // (LDS-resident iterate: owner-mapped passes)
        bool done = false;
        if constexpr (kLatOnly) { if (L.nx == 12 && L.nu == 4) { check_norms_own<12, 4>(c, Xl, Zl, Yl, D, E, S.Qv, cc, nrm, vsum, scaled); done = true; } }      // (mpcqp_w8.hip: the BASELINE shape unrolled)
        // (the reference's cart pole on the dense backend: its nx-long sums unrolled -- one workgroup per compute unit has the registers for it)
        if constexpr (!kLatOnly && OCC == 1 && NB == 16) { if (L.dense && L.nx == 4 && L.nu == 1) { check_norms_own<4, 1>(c, Xl, Zl, Yl, D, E, S.Qv, cc, nrm, vsum, scaled); done = true; } }
        if (!done) check_norms_own(c, Xl, Zl, Yl, D, E, S.Qv, cc, nrm, vsum, scaled);
    }
    else check_norms_gown(c, X, Z, Y, D, E, S.Qv, cc, S.T, S.tv, nrm, vsum, scaled);   // (iterate in global memory: staged, then the same passes)
    TICK(12)
    if (scaled) block_reduce<11, 1>(nrm, vsum, S.red);
    else block_reduce<7, 1>(nrm, vsum, S.red);
    obj_val = vsum[0]; pri_res = nrm[0]; dua_res = nrm[3];
    TICK(13)

    // ---- OSQP's infeasibility certificates (paper section 3.5) on the last increments, in unscaled terms
    auto primal_infeasible = [&](double eps) -> bool {
        // v = c * delta_y (= E * scaled delta_y), projected on the polar of the recession cone of [l,u]
        double vmax[1] = {0.0}, vs[1] = {0.0};
        for (int r = tid; r < L.m; r += NT) {
            double lo, hi; row_bounds(c, S.x0s, S.du0, r, lo, hi);
            double e = E[r], v = cc * dyg[r];
            if (e * hi > QP_INFTY * MIN_SCALING) { if (e * lo < -QP_INFTY * MIN_SCALING) v = 0.0; else v = fmin(v, 0.0); }
            else if (e * lo < -QP_INFTY * MIN_SCALING) v = fmax(v, 0.0);
            S.T[r] = v;
            vmax[0] = fmax(vmax[0], fabs(v));
            vs[0] += hi * fmax(v, 0.0) + lo * fmin(v, 0.0);
        }
        block_reduce<1, 1>(vmax, vs, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vs[0] < -eps * nd)) return false;
        double amax[1] = {0.0}, dummy[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double a = 0.0; AT_row(c, j, [&](double co, int row) { a += co * S.T[row]; });
            amax[0] = fmax(amax[0], fabs(a));
        }
        block_reduce<1, 1>(amax, dummy, S.red);
        return amax[0] < eps * nd;
    };
    auto dual_infeasible = [&](double eps) -> bool {
        double vmax[1] = {0.0}, vs[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double d = dxg[j];
            vmax[0] = fmax(vmax[0], fabs(d));
            vs[0] += ((j < L.oe) ? S.Qv[j] : 0.0) * d;
        }
        block_reduce<1, 1>(vmax, vs, S.red);
        double nd = vmax[0];
        if (!(nd > eps)) return false;
        if (!(vs[0] < -eps * nd)) return false;
        double pmax[1] = {0.0}, bad[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) {
            double a = 0.0; P_row(c, j, [&](double co, int idx) { a += co * dxg[idx]; });
            pmax[0] = fmax(pmax[0], fabs(a));
        }
        for (int r = tid; r < L.m; r += NT) {
            double a = 0.0; A_row(c, r, [&](double co, int idx) { a += co * dxg[idx]; });
            double lo, hi; row_bounds(c, S.x0s, S.du0, r, lo, hi);
            double e = E[r];
            if ((e * hi < QP_INFTY * MIN_SCALING && a > eps * nd) || (e * lo > -QP_INFTY * MIN_SCALING && a < -eps * nd)) bad[0] = 1.0;
        }
        block_reduce<1, 1>(pmax, bad, S.red);
        return (pmax[0] < eps * nd) && (bad[0] == 0.0);
    };
    auto check_termination = [&](bool approx) -> bool {
        double ea = S_.eps_abs, er = S_.eps_rel, epi = S_.eps_prim_inf, edi = S_.eps_dual_inf;
        // (fmax drops NaN operands, so a NaN iterate leaves both residuals at zero: the objective, a plain sum, carries it)
        if (pri_res > QP_INFTY || dua_res > QP_INFTY || obj_val != obj_val) { status = MPCQP_NON_CVX; obj_val = NAN; return true; }
        if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
        bool pc = pri_res < ea + er * fmax(nrm[2], nrm[1]);
        bool dc = dua_res < ea + er * fmax(fmax(nrm[6], nrm[5]), nrm[4]);
        bool pic = false, dic = false;
        if (!pc) pic = primal_infeasible(epi);
        if (!dc) dic = dual_infeasible(edi);
        if (pc && dc) { status = approx ? MPCQP_SOLVED_INACCURATE : MPCQP_SOLVED; return true; }
        if (pic) { status = approx ? MPCQP_PRIMAL_INFEASIBLE_INACCURATE : MPCQP_PRIMAL_INFEASIBLE; obj_val = QP_INFTY; return true; }
        if (dic) { status = approx ? MPCQP_DUAL_INFEASIBLE_INACCURATE : MPCQP_DUAL_INFEASIBLE; obj_val = -QP_INFTY; return true; }
        return false;
    };

    int term = 0, rho_upd = 0;
    if (mode & COLD_PLAIN) { status = MPCQP_UNSOLVED; term = 1; }
    else {
        if (mode & COLD_CHECK) term = check_termination(false) ? 1 : 0;
        if (!term && (mode & COLD_FINAL)) {             // iteration limit: OSQP retries with 10x looser tolerances
            if (!check_termination(true)) status = MPCQP_MAX_ITER_REACHED;
            term = 1;
        }
        if (!term && (mode & COLD_RHO)) {
            double pri = nrm[7] / (nrm[8] + 1e-10), dua = nrm[9] / (nrm[10] + 1e-10);
            double rn = fmin(fmax(rho * sqrt(pri / (dua + 1e-10)), RHO_MIN), RHO_MAX);
            if (rn > rho * S_.adaptive_rho_tolerance || rn < rho / S_.adaptive_rho_tolerance) {
                rho = rn;
                for (int r = tid; r < L.m; r += NT) om[r] = row_rho(ctp[r], rho) * E[r] * E[r];
                __syncthreads();
                { int pin = 0; asm volatile("" : "+v"(pin)); run_factor_phase<NB, OCC>(&pin); asm volatile("" :: "v"(pin)); }
                rho_upd = 1;
            }
        }
    }
    __syncthreads();
    TICK(14)
    if (term) {      // solution, and the iterate the next warm start begins from
        const bool has_sol = !(status == MPCQP_PRIMAL_INFEASIBLE || status == MPCQP_PRIMAL_INFEASIBLE_INACCURATE ||
                               status == MPCQP_DUAL_INFEASIBLE || status == MPCQP_DUAL_INFEASIBLE_INACCURATE || status == MPCQP_NON_CVX);
        double *xo = P.xo + (size_t)b * L.n, *yo = P.yo + (size_t)b * L.m;
        for (int j = tid; j < L.n; j += NT) { double v = gx[j]; xo[j] = has_sol ? v : NAN; if (!has_sol) gx[j] = 0.0; }
        for (int r = tid; r < L.m; r += NT) { double v = gy[r]; yo[r] = has_sol ? v : NAN; if (!has_sol) { gy[r] = 0.0; gz[r] = 0.0; } }
    }
    if (term && tid < L.nu) S.uo[tid] = X[L.ou + tid];                     // (the closed loop on the device: output() of the next step, with iflag[4])
    if (tid == 0) {
        mpcqp_info inf;                                                   // (written whole: the two running counts live in LDS, nothing is read back)
        inf.status = status; inf.iter = iter; inf.rho_updates = (S.iflag[1] += rho_upd); inf.reserved = (S.iflag[3] += 1);
        inf.obj_val = obj_val; inf.pri_res = pri_res; inf.dua_res = dua_res; inf.rho = rho;
        P.rho[b] = rho;
        S.iflag[4] = status;
        if (term) {
            atomicAdd(&P.stats[0], (unsigned long long)iter); atomicAdd(&P.stats[1], (unsigned long long)inf.reserved);
            atomicAdd(&P.stats[2], (unsigned long long)inf.rho_updates); atomicAdd(&P.stats[3], 1ULL);
            P.work[b] += (unsigned)iter;
            inf.reserved = 0;
        }
        P.info[b] = inf;
    }
    TICK(10)
    TICK_FLUSH
    return term;
}

// ---- hot-loop pieces.  NXT/NUT: compile-time nx/nu (0 = take them from the layout at run time).
#ifndef MPCQP_HOT_U
#define MPCQP_HOT_U 4
#endif
constexpr int HOT_U = MPCQP_HOT_U;                   // global-memory iterate: elements per thread whose loads are issued together
template <int NXT> __device__ __forceinline__ int hx(const Lay &L) { return NXT ? NXT : L.nx; }
template <int NUT> __device__ __forceinline__ int hu(const Lay &L) { return NUT ? NUT : L.nu; }
template <int NXT> __device__ __forceinline__ int divx(const Lay &L, int v) { return NXT ? v / NXT : idiv(v, L.rnx); }
template <int NUT> __device__ __forceinline__ int divu(const Lay &L, int v) { return NUT ? v / NUT : idiv(v, L.rnu); }

// ------------------------------------------------------------------------------------------------
// Steps (1)-(2) and (4)-(6) of the ADMM iteration -- the parallel phases around the KKT solve:
//   W = omega z - c y                       (rows; left behind by the previous update)
//   rhs = s x - c q + A' W                  (variables), with the slack elimination fused in:
//   te = rhs_eps / kappa -> W[box row]      Tc[k][a] = rhs_x - omega_box te  |  rhs_u  |  0 (padding)
//   [solve]  slack back-substitution, zt = A xt, relaxation, projection on [l,u], dual update, x update.
// For the LDS-resident iterate (small problems) with an OWNER map instead of flat row / variable loops: thread t owns, for j = 0, 1, the state element e = t + NT j = (stage k, component a) -- the variable x_k[a], its slack,
// the dynamics row e and the state-box row rs + e, which all share that index -- and the input element cu = t = (k, jj) -- the
// variable u_k[jj], its box row ri + cu, the Delta-u row rdu + nu + cu and, for cu < nu, the first-step row rdu + cu.
//   * every pass runs ONE kind of item (no divergent tree over four row types and two variable kinds per wave: the flat
//     loops spent most of their instructions on that tree, on index arithmetic and on scalar reloads inside it);
//   * what the thread needs of omega, s, c q sits in its registers for the round (15 doubles);
//   * slack back-substitution, relaxation and the row updates of an element happen in one pass (the slack value never leaves
//     the thread), so the barrier between "variables" and "rows" is gone;
//   * one division per row (c y / omega; omega / c is a multiplication by 1/c).
// Needs n_x <= 2 NT and n_u + nu <= NT (the host only chooses the LDS-resident mode then).  Layouts of X, Z, Y, W are unchanged.
// ------------------------------------------------------------------------------------------------
struct OwnRegs {
    double sv_x[2], cq_x[2], sv_e[2], om_s[2], om_d[2];      // per state element: s of x and of its slack, c q, omega of the box row / of the dynamics row
    double sv_u, cq_u, om_i, om_du, om_d0;                    // input element: s, c q, omega of the box row, of Delta-u row nu + cu, of first-step row cu
};
template <int NB, int NXT, int NUT>
__device__ __forceinline__ void own_load(const Lay &L, cgdouble *om, cgdouble *sv, cgdouble *qv, double cc, OwnRegs &h) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        const bool v = e < L.n_x;
        h.sv_x[j] = v ? sv[e] : 0.0; h.cq_x[j] = v ? cc * qv[e] : 0.0; h.sv_e[j] = (v && L.soft) ? sv[L.oe + e] : 0.0;
        h.om_s[j] = v ? om[L.rs + e] : 1.0; h.om_d[j] = v ? om[e] : 1.0;
    }
    const int cu = NT - 1 - tid;                         // (input elements from the last thread downwards: the first threads own two state elements)
    const bool v = cu < L.n_u;
    h.sv_u = v ? sv[L.ou + cu] : 0.0; h.cq_u = v ? cc * qv[L.n_x + cu] : 0.0;
    h.om_i = v ? om[L.ri + cu] : 1.0; h.om_du = v ? om[L.rdu + L.nu + cu] : 1.0; h.om_d0 = cu < L.nu ? om[L.rdu + cu] : 1.0;
}
// Inside a round the dual variable is carried as ys = c y / omega: the projection  z+ = clamp(zr + c y / omega)  and the dual step
// y+ = y + (omega / c)(zr - z+)  become  z+ = clamp(zr + ys),  ys+ = ys + (zr - z+),  W = omega (z+ - ys+)  -- no division per row
// and iteration.  Rows are converted when a round starts (own_rows_w / gown_rows_w) and back when it ends (own_finish /
// gown_finish); every other phase sees y itself.
__device__ __forceinline__ void own_scale_row(double w, double cc, const double *Z, double *Y, double *W, int r) {
    const double ys = cc * Y[r] / w;
    Y[r] = ys; W[r] = w * (Z[r] - ys);
}
// W = omega z - c y of the thread's rows (first iteration of a round; afterwards own_update leaves it behind); zero padding of Tc
template <int NB>
__device__ __forceinline__ void own_rows_w(const Lay &L, const OwnRegs &h, double cc, const double *Z, double *Y, double *W, double *Tc) {
    const int tid = opaque_lane(threadIdx.x);
    for (int i = tid; i < L.N * NB; i += NT) Tc[i] = 0.0;      // (padding and absent inputs stay zero through the solves: the factor has zero rows there)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (e < L.n_x) { own_scale_row(h.om_d[j], cc, Z, Y, W, e); own_scale_row(h.om_s[j], cc, Z, Y, W, L.rs + e); }
    }
    const int cu = NT - 1 - tid;
    if (cu < L.n_u) {
        const int ri = L.ri + cu, rd = L.rdu + L.nu + cu;
        own_scale_row(h.om_i, cc, Z, Y, W, ri); own_scale_row(h.om_du, cc, Z, Y, W, rd);
        if (cu < L.nu) own_scale_row(h.om_d0, cc, Z, Y, W, L.rdu + cu);
    }
    __syncthreads();
}
// Nc < Np: the last input u_{Nc-1} is held to the end of the horizon, so its column of A has entries in the dynamics rows of ALL later stages
// and its part of A'W is a sum over Np - Nc + 1 stages.  One thread adding them up serially was a third of an iteration of the reference's
// Kalman notebook (Np = 150, Nc = 75).  The per-stage terms Bd' W_s are formed in parallel, parked in the input slots of the stages that
// carry no input (zero otherwise, and zero again before the solve), and the owner of (Nc - 1, jj) adds Np - Nc + 1 LDS values in a fixed order.
template <int NB>
__device__ __forceinline__ void held_input_terms(const Lay &L, int nx, int nu, const double *Bd, const double *W, double *Tc) {
    const int npair = (L.Np - L.Nc + 1) * nu;
    for (int p = NT - 1 - (int)threadIdx.x; p < npair; p += NT) {      // (the last threads: the first ones carry the extra trip of the state items)
        const int so = p / nu, jj = p - so * nu, s = L.Nc + so;
        const double *w1 = W + s * nx;
        double t = 0.0;
        for (int r = 0; r < nx; ++r) t += Bd[r * nu + jj] * w1[r];
        Tc[s * NB + nx + jj] = t;
    }
}
// The owner of (Nc - 1, jj) used to add the Np - Nc + 1 parked terms itself, one LDS read after the other (76 dependent reads at the notebook
// shape: 2 us of a 15 us iteration, every other thread waiting at the barrier).  Now wave 0 sums them -- lane l takes stages Nc + l, Nc + l + 64, ...,
// then a fixed-order butterfly over the lanes -- and leaves the nu sums in `out` (LDS, nu doubles); the parked slots are zeroed on the way.
// All threads call (one barrier inside); the order of the additions is fixed, so results do not depend on timing.
// ADD: the sums go straight onto the held input's right-hand side slots (stage Nc - 1), which their owners filled before the caller's barrier -- the
// same last addition as the owner's own `+= out[jj]`, without the second hand-over.
template <int NB, bool ADD = false>
__device__ __forceinline__ void held_input_reduce(const Lay &L, int nx, int nu, double *Tc, double *out) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        for (int jj = 0; jj < nu; ++jj) {
            double a = 0.0;
            for (int s = L.Nc + lane; s <= L.Np; s += 64) { a += Tc[s * NB + nx + jj]; Tc[s * NB + nx + jj] = 0.0; }
            a = wave_reduce<false>(a);
            if (lane == 0) { if (ADD) Tc[(L.Nc - 1) * NB + nx + jj] += a; else out[jj] = a; }
        }
    }
    __syncthreads();
}

// rhs = s x - c q + A' W with the slack eliminated, into Tc
template <int NB, int NXT, int NUT>
__device__ __forceinline__ void own_rhs(const Lay &L, const double *hot, const OwnRegs &h, double cc, const double *X, double *W, double *Tc, double *hsum) {
    const int tid = opaque_lane(threadIdx.x);        // (keeps the per-thread index arithmetic out of LICM's reach: hoisted, it spills)
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    const double cef = cc * hot[L.oeps];
    const bool held = L.Nc < L.Np;
    if (held) held_input_terms<NB>(L, nx, nu, Bd, W, Tc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (e < L.n_x) {
            const int k = divx<NXT>(L, e), a = e - k * nx;
            double rx = h.sv_x[j] * X[e] - h.cq_x[j] - W[e];
            if (k < L.Np) {
                const double *w1 = W + (k + 1) * nx;
#pragma unroll
                for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) rx += Ad[r * nx + a] * w1[r];
                if (!NXT) for (int r = 0; r < nx; ++r) rx += Ad[r * nx + a] * w1[r];
            }
            const double wsoft = W[L.rs + e];
            const double te = L.soft ? (h.sv_e[j] * X[L.oe + e] + wsoft) / (cef + h.sv_e[j] + h.om_s[j]) : 0.0;      // (hard box: no slack to eliminate)
            W[L.rs + e] = te;                          // read back by this thread in own_update
            Tc[k * NB + a] = rx + wsoft - h.om_s[j] * te;
        }
    }
    if (NT - 1 - tid < L.n_u) {
        const int cu = NT - 1 - tid, k = divu<NUT>(L, cu), jj = cu - k * nu;
        double ru = h.sv_u * X[L.ou + cu] - h.cq_u + W[L.ri + cu] - W[L.rdu + nu + cu];
        if (k == 0) ru += W[L.rdu + jj];
        if (cu > 0) ru += W[L.rdu + nu + cu - 1];
        if (!(held && k == L.Nc - 1)) {                                                   // (the held input's A'W: summed over the later stages below)
            const double *w1 = W + (k + 1) * nx;
#pragma unroll
            for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) ru += Bd[r * nu + jj] * w1[r];
            if (!NXT) for (int r = 0; r < nx; ++r) ru += Bd[r * nu + jj] * w1[r];
        }
        Tc[k * NB + nx + jj] = ru;
    }
    if (held) { __syncthreads(); held_input_reduce<NB, true>(L, nx, nu, Tc, hsum); }      // the last input acts on every later stage (mpc.py:540-543)
    else __syncthreads();
}
// slack back-substitution, zt = A xt, relaxation, projection on [l,u], dual update, x update
template <int NB, int NXT, int NUT>
__device__ __forceinline__ void own_update(const Lay &L, const double *hot, const double *x0s, const double *du0, const OwnRegs &h, double cc, double alpha,
                                           double *X, double *Z, double *Y, double *W, const double *Tc, bool keep_delta, gdouble *dxg, gdouble *dyg) {
    const int tid = opaque_lane(threadIdx.x);
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    const double cef = cc * hot[L.oeps], cinv = 1.0 / cc, beta = 1.0 - alpha;
    auto row = [&](int r, double w, double zt, double lo, double hi) {
        lo = lo < -QP_INFTY ? -QP_INFTY : lo;
        hi = hi > QP_INFTY ? QP_INFTY : hi;
        const double zv = Z[r], ys = Y[r];            // (ys = c y / omega, see own_scale_row)
        const double zr = alpha * zt + beta * zv;
        const double zn = fmin(fmax(zr + ys, lo), hi);
        const double d = zr - zn, ysn = ys + d;
        Z[r] = zn; Y[r] = ysn; W[r] = w * (zn - ysn);
        if (keep_delta) dyg[r] = (w * cinv) * d;
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (e < L.n_x) {
            const int k = divx<NXT>(L, e), a = e - k * nx;
            const double xt = Tc[k * NB + a];
            const double et = L.soft ? W[L.rs + e] - (h.om_s[j] / (cef + h.sv_e[j] + h.om_s[j])) * xt : 0.0;
            const double xo = X[e], xn = alpha * xt + beta * xo;
            X[e] = xn;
            if (keep_delta) dxg[e] = xn - xo;
            if (L.soft) { const double eo = X[L.oe + e], en = alpha * et + beta * eo; X[L.oe + e] = en; if (keep_delta) dxg[L.oe + e] = en - eo; }
            double zt = -xt;                           // dynamics row e
            if (k > 0) {
                const double *xp = Tc + (k - 1) * NB;
                const double *up = Tc + min(k - 1, L.Nc - 1) * NB + nx;
#pragma unroll
                for (int i = 0; i < (NXT ? NXT : 1); ++i) if (NXT) zt += Ad[a * nx + i] * xp[i];
                if (!NXT) for (int i = 0; i < nx; ++i) zt += Ad[a * nx + i] * xp[i];
#pragma unroll
                for (int i = 0; i < (NUT ? NUT : 1); ++i) if (NUT) zt += Bd[a * nu + i] * up[i];
                if (!NUT) for (int i = 0; i < nu; ++i) zt += Bd[a * nu + i] * up[i];
            }
            const double b0 = k == 0 ? -x0s[a] : 0.0;
            row(e, h.om_d[j], zt, b0, b0);
            row(L.rs + e, h.om_s[j], xt + et, hot[L.oxmin + a], hot[L.oxmax + a]);      // state-box row (soft: x + eps)
        }
    }
    if (NT - 1 - tid < L.n_u) {
        const int cu = NT - 1 - tid, k = divu<NUT>(L, cu), jj = cu - k * nu;
        const double ut = Tc[k * NB + nx + jj];
        const double uo = X[L.ou + cu], un = alpha * ut + beta * uo;
        X[L.ou + cu] = un;
        if (keep_delta) dxg[L.ou + cu] = un - uo;
        row(L.ri + cu, h.om_i, ut, hot[L.oumin + jj], hot[L.oumax + jj]);
        double zt = -ut;                               // Delta-u row nu + cu: next flattened input minus this one (mpc.py:570)
        if (cu + 1 < L.n_u) zt += (jj + 1 < nu) ? Tc[k * NB + nx + jj + 1] : Tc[(k + 1) * NB + nx];
        row(L.rdu + nu + cu, h.om_du, zt, hot[L.oDumin + jj], hot[L.oDumax + jj]);
        if (cu < nu) row(L.rdu + cu, h.om_d0, ut, du0[cu], du0[nu + cu]);              // first step: u_0 - u_{-1}
    }
    __syncthreads();
}

// end of a round: the thread's rows back from ys = c y / omega to y
__device__ __forceinline__ void own_finish(const Lay &L, const OwnRegs &h, double cc, double *Y) {
    const int tid = opaque_lane(threadIdx.x);
    const double cinv = 1.0 / cc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + NT * j;
        if (e < L.n_x) { Y[e] *= h.om_d[j] * cinv; Y[L.rs + e] *= h.om_s[j] * cinv; }
    }
    const int cu = NT - 1 - tid;
    if (cu < L.n_u) {
        Y[L.ri + cu] *= h.om_i * cinv; Y[L.rdu + L.nu + cu] *= h.om_du * cinv;
        if (cu < L.nu) Y[L.rdu + cu] *= h.om_d0 * cinv;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// The owner map for problems whose iterate does not fit LDS (x, z, y, omega, s in global memory): the same items, as many
// passes as it takes (state element e = t + NT j, input element cu = t + NT j), HOT_U items per thread at a time with ALL their
// global loads issued first -- through pointers the compiler knows to be global (a generic pointer means FLAT loads, which it
// may not reorder with the LDS stores of the item before) -- so a pass costs one memory round trip, not one per operand.
// Against the flat loops this replaces (one over the padded variables, one over the variables again, one over the m rows with
// a four-way branch per row): a third of the round trips and no divergent tree.
// ------------------------------------------------------------------------------------------------
template <int NB> struct GownCfg { static constexpr int U = 2; };
// INL: the iterate and the metric vectors are LDS copies staged for the round (admm_body: at most one instance per compute unit, so the copies
// fit) -- the same passes without the global round trips.  The pointer types say which: explicit global address space, or plain (LDS, inferred).
template <bool INL> struct GPtr { typedef cgdouble c; typedef gdouble m; };
template <> struct GPtr<true> { typedef const double c; typedef double m; };      // items per thread in flight (ten operands per input item; 2 / 4 / 8 measured at cfg-5: 95.0 / 94.2 / 92.2 k solves/s)
template <int NB, bool INL = false>
__device__ __forceinline__ void gown_rows_w(const Lay &L, typename GPtr<INL>::c *om, double cc, const double *Z, double *Y, double *W, double *Tc) {
    typedef typename GPtr<INL>::c cgdouble; typedef typename GPtr<INL>::m gdouble;
    const int tid = opaque_lane(threadIdx.x);
    cgdouble *Zg = (cgdouble *)Z, *Yg = (cgdouble *)Y;
    for (int i = tid; i < L.N * NB; i += NT) Tc[i] = 0.0;      // (padding and absent inputs stay zero through the solves: the factor has zero rows there)
#pragma unroll HOT_U
    for (int r = tid; r < L.m; r += NT) { const double w = om[r], ys = cc * Yg[r] / w; ((gdouble *)Y)[r] = ys; W[r] = w * (Zg[r] - ys); }      // (y -> c y / omega, see own_scale_row)
    __syncthreads();
}
// end of a round: y back from c y / omega
template <bool INL = false>
__device__ __forceinline__ void gown_finish(const Lay &L, typename GPtr<INL>::c *om, double cc, double *Y) {
    typedef typename GPtr<INL>::m gdouble;
    const int tid = opaque_lane(threadIdx.x);
    const double cinv = 1.0 / cc;
    gdouble *Yg = (gdouble *)Y;
#pragma unroll HOT_U
    for (int r = tid; r < L.m; r += NT) Yg[r] = Yg[r] * (om[r] * cinv);
    __syncthreads();
}
template <int NB, int NXT, int NUT, bool INL = false>
__device__ __forceinline__ void gown_rhs(const Lay &L, const double *hot, typename GPtr<INL>::c *om, typename GPtr<INL>::c *sv, typename GPtr<INL>::c *qv, double cc, const double *X, double *W, double *Tc, double *hsum) {
    typedef typename GPtr<INL>::c cgdouble;
    constexpr int GU = GownCfg<NB>::U;
    const int tid = opaque_lane(threadIdx.x);
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    const double cef = cc * hot[L.oeps];
    cgdouble *Xg = (cgdouble *)X;
    const bool held = L.Nc < L.Np;
    if (held) held_input_terms<NB>(L, nx, nu, Bd, W, Tc);
    for (int e0 = tid; e0 < L.n_x; e0 += GU * NT) {
        double svx[GU], qx[GU], oms[GU], xv[GU], sve[GU], xe[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = min(e0 + u * NT, L.n_x - 1);
            svx[u] = sv[e]; qx[u] = qv[e]; oms[u] = om[L.rs + e]; xv[u] = Xg[e]; sve[u] = 0.0; xe[u] = 0.0;
            if (L.soft) { sve[u] = sv[L.oe + e]; xe[u] = Xg[L.oe + e]; }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = e0 + u * NT;
            if (e < L.n_x) {
                const int k = divx<NXT>(L, e), a = e - k * nx;
                double rx = svx[u] * xv[u] - cc * qx[u] - W[e];
                if (k < L.Np) {
                    const double *w1 = W + (k + 1) * nx;
#pragma unroll
                    for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) rx += Ad[r * nx + a] * w1[r];
                    if (!NXT) for (int r = 0; r < nx; ++r) rx += Ad[r * nx + a] * w1[r];
                }
                const double wsoft = W[L.rs + e];
                const double te = L.soft ? (sve[u] * xe[u] + wsoft) / (cef + sve[u] + oms[u]) : 0.0;      // (hard box: no slack to eliminate)
                W[L.rs + e] = te;                      // read back by this thread in gown_update
                Tc[k * NB + a] = rx + wsoft - oms[u] * te;
            }
        }
    }
    // (input items from the LAST thread downwards: the threads that ran the short last trip of the state items above are the first ones, so the
    //  waves' item counts differ by at most one -- with one instance alone on the compute unit the pass lasts as long as its busiest wave)
    for (int c0 = NT - 1 - tid; c0 < L.n_u; c0 += GU * NT) {
        double svu[GU], qu[GU], uv[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) { const int cu = min(c0 + u * NT, L.n_u - 1); svu[u] = sv[L.ou + cu]; qu[u] = qv[L.n_x + cu]; uv[u] = Xg[L.ou + cu]; }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int cu = c0 + u * NT;
            if (cu < L.n_u) {
                const int k = divu<NUT>(L, cu), jj = cu - k * nu;
                double ru = svu[u] * uv[u] - cc * qu[u] + W[L.ri + cu] - W[L.rdu + nu + cu];
                if (k == 0) ru += W[L.rdu + jj];
                if (cu > 0) ru += W[L.rdu + nu + cu - 1];
                if (!(held && k == L.Nc - 1)) {                                           // (the held input's A'W: summed over the later stages below)
                    const double *w1 = W + (k + 1) * nx;
#pragma unroll
                    for (int r = 0; r < (NXT ? NXT : 1); ++r) if (NXT) ru += Bd[r * nu + jj] * w1[r];
                    if (!NXT) for (int r = 0; r < nx; ++r) ru += Bd[r * nu + jj] * w1[r];
                }
                Tc[k * NB + nx + jj] = ru;
            }
        }
    }
    if (held) { __syncthreads(); held_input_reduce<NB, true>(L, nx, nu, Tc, hsum); }      // the last input acts on every later stage (mpc.py:540-543)
    else __syncthreads();
}
template <int NB, int NXT, int NUT, bool INL = false>
__device__ __forceinline__ void gown_update(const Lay &L, const double *hot, const double *x0s, const double *du0, typename GPtr<INL>::c *om, typename GPtr<INL>::c *sv, double cc, double alpha,
                                            double *X, double *Z, double *Y, double *W, const double *Tc, bool keep_delta, gdouble_g *dxg, gdouble_g *dyg) {
    typedef typename GPtr<INL>::m gdouble;
    constexpr int GU = GownCfg<NB>::U;
    const int tid = opaque_lane(threadIdx.x);
    const int nx = hx<NXT>(L), nu = hu<NUT>(L);
    const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
    const double cef = cc * hot[L.oeps], cinv = 1.0 / cc, beta = 1.0 - alpha;
    gdouble *Xg = (gdouble *)X, *Zg = (gdouble *)Z, *Yg = (gdouble *)Y;
    // one row: relaxation, projection, dual step; z, y come in as loaded and leave through global stores
    auto row = [&](int r, double w, double zt, double lo, double hi, double zv, double ys) {      // (ys = c y / omega, see own_scale_row)
        lo = lo < -QP_INFTY ? -QP_INFTY : lo;
        hi = hi > QP_INFTY ? QP_INFTY : hi;
        const double zr = alpha * zt + beta * zv;
        const double zn = fmin(fmax(zr + ys, lo), hi);
        const double d = zr - zn, ysn = ys + d;
        Zg[r] = zn; Yg[r] = ysn; W[r] = w * (zn - ysn);
        if (keep_delta) dyg[r] = (w * cinv) * d;
    };
    for (int e0 = tid; e0 < L.n_x; e0 += GU * NT) {
        double oms[GU], omd[GU], sve[GU], xo[GU], eo[GU], zd[GU], yd[GU], zs[GU], ys[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = min(e0 + u * NT, L.n_x - 1);
            oms[u] = om[L.rs + e]; omd[u] = om[e]; xo[u] = Xg[e]; zd[u] = Zg[e]; yd[u] = Yg[e]; zs[u] = Zg[L.rs + e]; ys[u] = Yg[L.rs + e];
            sve[u] = 0.0; eo[u] = 0.0;
            if (L.soft) { sve[u] = sv[L.oe + e]; eo[u] = Xg[L.oe + e]; }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = e0 + u * NT;
            if (e < L.n_x) {
                const int k = divx<NXT>(L, e), a = e - k * nx;
                const double xt = Tc[k * NB + a];
                const double et = L.soft ? W[L.rs + e] - (oms[u] / (cef + sve[u] + oms[u])) * xt : 0.0;
                const double xn = alpha * xt + beta * xo[u];
                Xg[e] = xn;
                if (keep_delta) dxg[e] = xn - xo[u];
                if (L.soft) { const double en = alpha * et + beta * eo[u]; Xg[L.oe + e] = en; if (keep_delta) dxg[L.oe + e] = en - eo[u]; }
                double zt = -xt;                       // dynamics row e
                if (k > 0) {
                    const double *xp = Tc + (k - 1) * NB;
                    const double *up = Tc + min(k - 1, L.Nc - 1) * NB + nx;
#pragma unroll
                    for (int i = 0; i < (NXT ? NXT : 1); ++i) if (NXT) zt += Ad[a * nx + i] * xp[i];
                    if (!NXT) for (int i = 0; i < nx; ++i) zt += Ad[a * nx + i] * xp[i];
#pragma unroll
                    for (int i = 0; i < (NUT ? NUT : 1); ++i) if (NUT) zt += Bd[a * nu + i] * up[i];
                    if (!NUT) for (int i = 0; i < nu; ++i) zt += Bd[a * nu + i] * up[i];
                }
                const double b0 = k == 0 ? -x0s[a] : 0.0;
                row(e, omd[u], zt, b0, b0, zd[u], yd[u]);
                row(L.rs + e, oms[u], xt + et, hot[L.oxmin + a], hot[L.oxmax + a], zs[u], ys[u]);      // state-box row (soft: x + eps)
            }
        }
    }
    for (int c0 = NT - 1 - tid; c0 < L.n_u; c0 += GU * NT) {      // (from the last thread downwards, as in gown_rhs)
        double uo[GU], omi[GU], omu[GU], om0[GU], zi[GU], yi[GU], zu[GU], yu[GU], z0[GU], y0[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int cu = min(c0 + u * NT, L.n_u - 1), r0 = L.rdu + min(cu, nu - 1);
            uo[u] = Xg[L.ou + cu]; omi[u] = om[L.ri + cu]; omu[u] = om[L.rdu + nu + cu]; zi[u] = Zg[L.ri + cu]; yi[u] = Yg[L.ri + cu];
            zu[u] = Zg[L.rdu + nu + cu]; yu[u] = Yg[L.rdu + nu + cu]; om0[u] = om[r0]; z0[u] = Zg[r0]; y0[u] = Yg[r0];
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int cu = c0 + u * NT;
            if (cu < L.n_u) {
                const int k = divu<NUT>(L, cu), jj = cu - k * nu;
                const double ut = Tc[k * NB + nx + jj];
                const double un = alpha * ut + beta * uo[u];
                Xg[L.ou + cu] = un;
                if (keep_delta) dxg[L.ou + cu] = un - uo[u];
                row(L.ri + cu, omi[u], ut, hot[L.oumin + jj], hot[L.oumax + jj], zi[u], yi[u]);
                double zt = -ut;                       // Delta-u row nu + cu: next flattened input minus this one (mpc.py:570)
                if (cu + 1 < L.n_u) zt += (jj + 1 < nu) ? Tc[k * NB + nx + jj + 1] : Tc[(k + 1) * NB + nx];
                row(L.rdu + nu + cu, omu[u], zt, hot[L.oDumin + jj], hot[L.oDumax + jj], zu[u], yu[u]);
                if (cu < nu) row(L.rdu + cu, om0[u], ut, du0[cu], du0[nu + cu], z0[u], y0[u]);      // first step: u_0 - u_{-1}
            }
        }
    }
    __syncthreads();
}

// `iters` ADMM iterations of this workgroup's instance.  Expects the hot model prefix and the step data in LDS
// (load_common) and, with LDSSTATE, X/Z/Y carved behind the common area.
// MODE: how the KKT system is solved -- block-tridiagonal sweeps (MODE_CHAIN), the same with the bordered correction of a
// control horizon Nc < Np (MODE_BORDER), or the dense register-resident inverse of small problems (MODE_DENSE, mpcqp_dense.h).
// MODE_BCR + N: block cyclic reduction with the factor of an N-stage problem resident in registers (mpcqp_bcr.h).
// MODE_BCRT + N: the same with a dense top instead of the levels above 1, for any number of waves per workgroup (mpcqp_latw.h).
enum { MODE_CHAIN = 0, MODE_BORDER = 1, MODE_DENSE = 2, MODE_BCR = 100, MODE_BCRT = 200 };
template <int NB, int MAXB> __device__ __forceinline__ void admm_tiny(const Lay &, const HotPtrs &, Smem &, double *, double *, double *, double, int);      // mpcqp_tiny.h
template <int NXT, int NUT, int NST> __device__ __forceinline__ void admm_lat(const Lay &, const HotPtrs &, Smem &, double *, double *, double *, double, int);      // mpcqp_lat.h
template <int NXT, int NUT, int NST> __device__ __forceinline__ int admm_latw(const Lay &, const HotPtrs &, Smem &, double *, double *, double *, double, int, int);     // mpcqp_latw.h
// The ADMM round of a problem whose iterate does not live in LDS for the owner-mapped phases above (n_x > 2 NT, or too large for four workgroups
// per CU).  INL = false: x, z, y, omega, s, q in global memory (L2 / HBM), every pass pays its round trips -- right for batches, where other
// workgroups fill them.  INL = true (the handle holds at most one instance per compute unit, Lay::lstage): everything the passes read is STAGED
// into LDS behind the work area for the round -- iterate, metric vectors, linear cost and, with a held input, the two border matrices -- and the
// iterate goes back when the round ends: the same code without a global access inside an iteration (a lone long-horizon controller spent
// more than half of its iteration in those round trips).
template <int NB, int NXT, int NUT, int MODE, bool INL>
__device__ __forceinline__ void admm_round_global(const Lay &L, const HotPtrs &P, Smem &S, double alpha, int iters) {
    typedef typename GPtr<INL>::c cptr;
    const int b = inst_of(P.perm), tid = threadIdx.x;
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    double *W = S.T, *Tc = S.T + L.m;
    constexpr bool BORDER = MODE == MODE_BORDER;
    const size_t npb = (size_t)L.nu * L.N * L.NB;
    const double *Bg = BORDER ? P.Bb + b * npb : nullptr, *Zg = BORDER ? P.Zb + b * npb : nullptr;
    double *X = gx, *Z = gz, *Y = gy;
    const double *omp = P.omega + (size_t)b * L.m, *svp = P.s + (size_t)b * L.n, *qvp = S.Qv, *Bb = Bg, *Zb = Zg, *Sg = BORDER ? P.Sig + (size_t)b * L.nu * L.nu : nullptr;
    if constexpr (INL) {
        double *p = S.T + L.tsz;                      // [ x | z | y | omega | s | q | Bb | Zb ] behind the work area
        const int nq = L.n_x + L.n_u;
        double *Xs = carve(p, L.n), *Zs = carve(p, L.m), *Ys = carve(p, L.m), *oms = carve(p, L.m), *svs = carve(p, L.n), *qvs = carve(p, nq);
        for (int j = tid; j < L.n; j += NT) { Xs[j] = gx[j]; svs[j] = svp[j]; }
        for (int r = tid; r < L.m; r += NT) { Zs[r] = gz[r]; Ys[r] = gy[r]; oms[r] = omp[r]; }
        for (int j = tid; j < nq; j += NT) qvs[j] = qvp[j];
        X = Xs; Z = Zs; Y = Ys; omp = oms; svp = svs; qvp = qvs;
        if (BORDER) {
            double *Bs = carve(p, (int)npb), *Zbs = carve(p, (int)npb), *Sgs = carve(p, L.nu * L.nu);
            for (int i = tid; i < (int)npb; i += NT) { Bs[i] = Bg[i]; Zbs[i] = Zg[i]; }
            for (int i = tid; i < L.nu * L.nu; i += NT) Sgs[i] = P.Sig[(size_t)b * L.nu * L.nu + i];
            Bb = Bs; Zb = Zbs; Sg = Sgs;
        }
    }
    __syncthreads();
    cptr *gom = (cptr *)omp, *gsv = (cptr *)svp, *gqv = (cptr *)qvp;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double *F = factor_of(P, b);
    const double cc = P.c[b];
#ifndef MPCQP_ABL_NOPAR
    gown_rows_w<NB, INL>(L, gom, cc, Z, Y, W, Tc);
#endif
    TICK_RESET
    const int pace = INL ? 0 : __builtin_amdgcn_readfirstlane(pace_of(P.perm));
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;         // the increments feed the infeasibility certificates of the check
        // (all instances of the launch are resident at once and the launch ends with its slowest one: an instance expected to need few iterations
        //  idles here, leaving its share of the memory system to the stragglers -- rebalance() in mpcqp.hip sets the pace, results do not depend on it)
        for (int p = 0; p < pace; ++p) __builtin_amdgcn_s_sleep(127);
        TICK_START
#ifndef MPCQP_ABL_NOPAR
        gown_rhs<NB, NXT, NUT, INL>(L, S.hot, gom, gsv, gqv, cc, X, W, Tc, S.tv);
#endif
        TICK(0)
        if (BORDER) border_pre<NB>(L, Bb, Zb, Sg, Tc, S.tv, S.red);
        TICK(4)
        kkt_core<NB, NB == 16 && (NXT == 0 || NXT == 4)>(core_args(L, opaque_ptr(F), opaque_ptr((const double *)P.omega + (size_t)b * L.m)), Tc);
        if (BORDER) border_post(L, NB, Tc, S.tv);
#ifndef MPCQP_ABL_NOPAR
        gown_update<NB, NXT, NUT, INL>(L, S.hot, S.x0s, S.du0, gom, gsv, cc, alpha, X, Z, Y, W, Tc, keep_delta, dxg, dyg);
#endif
        TICK(5)
    }
    TICK_FLUSH
#ifndef MPCQP_ABL_NOPAR
    gown_finish<INL>(L, gom, cc, Y);
#endif
    if constexpr (INL) {
        for (int j = tid; j < L.n; j += NT) gx[j] = X[j];
        for (int r = tid; r < L.m; r += NT) { gz[r] = Z[r]; gy[r] = Y[r]; }
    }
}

template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE>
__device__ __forceinline__ void admm_body(const Lay &L, const HotPtrs &P, Smem &S, double *X, double *Z, double *Y, double alpha, int iters) {
    if constexpr (MODE == MODE_DENSE) { if (L.nb <= 8) admm_tiny<NB, 8>(L, P, S, X, Z, Y, alpha, iters); else admm_tiny<NB, 16>(L, P, S, X, Z, Y, alpha, iters); return; }      // register-resident iterate and inverse
    if constexpr (MODE >= MODE_BCRT) { admm_latw<NXT, NUT, MODE - MODE_BCRT>(L, P, S, X, Z, Y, alpha, iters, -1); return; }   // ... and cyclic-reduction factor with a dense top (k_mpc_run calls it directly, with its own termination test)
    else if constexpr (MODE >= MODE_BCR) { admm_lat<NXT, NUT, MODE - MODE_BCR>(L, P, S, X, Z, Y, alpha, iters); return; } // ... and cyclic-reduction factor
    if constexpr (!LDSSTATE) {          // iterate in global memory -- or staged in LDS for the round (generic kernels of small batches)
        if constexpr (NXT == 0 || NXT == 4) { if (L.lstage) { admm_round_global<NB, NXT, NUT, MODE, true>(L, P, S, alpha, iters); return; } }
        admm_round_global<NB, NXT, NUT, MODE, false>(L, P, S, alpha, iters);
        return;
    }
    // small-problem mode: the iterate x, z, y lives in LDS for the whole round, the owner-mapped phases above
    const int b = inst_of(P.perm), tid = threadIdx.x;
    double *gx = P.x + (size_t)b * L.n, *gz = P.z + (size_t)b * L.m, *gy = P.y + (size_t)b * L.m;
    double *W = S.T, *Tc = S.T + L.m;
    for (int j = tid; j < L.n; j += NT) X[j] = gx[j];
    for (int r = tid; r < L.m; r += NT) { Z[r] = gz[r]; Y[r] = gy[r]; }
    __syncthreads();
    cgdouble *gom = (cgdouble *)(P.omega + (size_t)b * L.m), *gsv = (cgdouble *)(P.s + (size_t)b * L.n), *gqv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double *F = factor_of(P, b);
    const double cc = P.c[b];
    constexpr bool BORDER = MODE == MODE_BORDER;
    OwnRegs hr;
    own_load<NB, NXT, NUT>(L, gom, gsv, gqv, cc, hr);
#ifndef MPCQP_ABL_NOPAR
    own_rows_w<NB>(L, hr, cc, Z, Y, W, Tc);
#endif
    TICK_RESET
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;         // the increments feed the infeasibility certificates of the check
        TICK_START
#ifndef MPCQP_ABL_NOPAR
        own_rhs<NB, NXT, NUT>(L, S.hot, hr, cc, X, W, Tc, S.tv);
#endif
        TICK(0)
        BorderPtrs bp; bp.red = S.red;
        if (BORDER) {
            const size_t npb = (size_t)L.nu * L.N * L.NB;
            bp.Bb = (double *)P.Bb + b * npb; bp.Zb = (double *)P.Zb + b * npb; bp.Sig = (double *)P.Sig + (size_t)b * L.nu * L.nu;
        }
        if (BORDER) border_pre<NB>(L, bp.Bb, bp.Zb, bp.Sig, Tc, S.tv, S.red);
        kkt_core<NB, NB == 16 && (NXT == 0 || NXT == 4)>(core_args(L, opaque_ptr(F), opaque_ptr((const double *)P.omega + (size_t)b * L.m)), Tc);
        if (BORDER) border_post(L, NB, Tc, S.tv);
#ifndef MPCQP_ABL_NOPAR
        own_update<NB, NXT, NUT>(L, S.hot, S.x0s, S.du0, hr, cc, alpha, X, Z, Y, W, Tc, keep_delta, dxg, dyg);
#endif
        TICK(5)
    }
    TICK_FLUSH
#ifndef MPCQP_ABL_NOPAR
    own_finish(L, hr, cc, Y);
#endif
    for (int j = tid; j < L.n; j += NT) gx[j] = X[j];
    for (int r = tid; r < L.m; r += NT) { gz[r] = Z[r]; gy[r] = Y[r]; }
}

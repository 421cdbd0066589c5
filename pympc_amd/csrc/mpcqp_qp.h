// mpcqp_qp.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The reference's QP, matrix-free: row visitors of A, A', P (mpc.py:482-598), bounds, linear cost, block reductions,
// entries of the reduced KKT matrix.
#pragma once

// iu Qu + dk QDu exactly as the reference forms it (mpc.py:510-526: two products, each rounded, then the sum): the entry of P must be BIT-identical
// to scipy's.  __dmul_rn / __dadd_rn do not stop the compiler from contracting the pair into one fused multiply-add (seen with iu = 5: 0.6000000000000001
// instead of 0.6; a power-of-two iu hides it), so the products are pinned in registers first.
__device__ __forceinline__ double input_weight(double iu, double qu, double dk, double qdu) {
    double a = iu * qu, b = dk * qdu;
    asm volatile("" : "+v"(a), "+v"(b));
    return a + b;
}


// ------------------------------------------------------------------------------------------------
// Row visitors: enumerate (coefficient, index) of one row of A, one column of A, one row of P.
// They ARE the device-side definition of the reference's matrices (mpc.py:482-598).
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void A_row(const Ctx &c, int r, F f) {
    const Lay &L = c.L;
    if (r < L.rs) {                                   // dynamics rows  (mpc.py:537-552)
        int k = idiv(r, L.rnx), i = r - k * L.nx;
        f(-1.0, r);
        if (k > 0) {
            const double *a = c.Ad() + i * L.nx;
            int base = (k - 1) * L.nx;
            for (int j = 0; j < L.nx; ++j) f(a[j], base + j);
            int ku = min(k - 1, L.Nc - 1);
            const double *b = c.Bd() + i * L.nu;
            base = L.ou + ku * L.nu;
            for (int j = 0; j < L.nu; ++j) f(b[j], base + j);
        }
    } else if (r < L.ri) {                            // soft state box: x_k + eps_k  (mpc.py:555-559)
        int j = r - L.rs;
        f(1.0, j);
        if (L.soft) f(1.0, L.oe + j);             // (SOFT_ON = False: the box is on x_k itself, mpc.py:555-557 without the eps block)
    } else if (r < L.rdu) {                           // input box  (mpc.py:561-565)
        f(1.0, L.ou + (r - L.ri));
    } else {                                          // Delta-u rows  (mpc.py:569-580)
        int rr = r - L.rdu;
        if (rr < L.nu) f(1.0, L.ou + rr);
        else {
            int cc = rr - L.nu;                       // -I + superdiagonal at offset ONE SCALAR
            f(-1.0, L.ou + cc);
            if (cc + 1 < L.n_u) f(1.0, L.ou + cc + 1);
        }
    }
}

template <class F>
__device__ __forceinline__ void AT_row(const Ctx &c, int j, F f) {     // column j of A
    const Lay &L = c.L;
    if (j < L.ou) {
        int k = idiv(j, L.rnx), i = j - k * L.nx;
        f(-1.0, j);
        if (k < L.Np) {
            const double *a = c.Ad() + i;
            int base = (k + 1) * L.nx;
            for (int r = 0; r < L.nx; ++r) f(a[r * L.nx], base + r);
        }
        f(1.0, L.rs + j);
    } else if (j < L.oe) {
        int cc = j - L.ou;
        int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
        int s_end = (k == L.Nc - 1) ? L.Np : k + 1;   // the last input is held to the end of the horizon
        const double *b = c.Bd() + jj;
        for (int s = k + 1; s <= s_end; ++s) {
            int base = s * L.nx;
            for (int r = 0; r < L.nx; ++r) f(b[r * L.nu], base + r);
        }
        f(1.0, L.ri + cc);
        if (k == 0) f(1.0, L.rdu + jj);
        f(-1.0, L.rdu + L.nu + cc);
        if (cc > 0) f(1.0, L.rdu + L.nu + cc - 1);
    } else {
        f(1.0, L.rs + (j - L.oe));
    }
}

// P as the solver sees it: the upper triangle of the reference's P mirrored (osqp keeps triu(P)).
template <class F>
__device__ __forceinline__ void P_row(const Ctx &c, int j, F f) {
    const Lay &L = c.L;
    if (j < L.ou) {                                    // blkdiag(I (x) Qx, QxN)  (mpc.py:486-487)
        int k = idiv(j, L.rnx), i = j - k * L.nx;
        const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
        int base = k * L.nx;
        for (int l = 0; l < L.nx; ++l) f(Q[min(i, l) * L.nx + max(i, l)], base + l);
    } else if (j < L.oe) {                             // diag(iU) (x) Qu + iDu (x) QDu  (mpc.py:505-526)
        int cc = j - L.ou;
        int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
        double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
        double dk = (k == L.Nc - 1) ? 1.0 : 2.0;
        const double *Qu = c.Qu(), *QDu = c.QDu();
        int base = L.ou + k * L.nu;
        for (int l = 0; l < L.nu; ++l) {
            int a = min(jj, l), b = max(jj, l);
            f(input_weight(iu, Qu[a * L.nu + b], dk, QDu[a * L.nu + b]), base + l);
        }
        if (k + 1 < L.Nc) for (int l = 0; l < L.nu; ++l) f(-QDu[jj * L.nu + l], base + L.nu + l);
        if (k > 0) for (int l = 0; l < L.nu; ++l) f(-QDu[l * L.nu + jj], base - L.nu + l);
    } else {
        f(c.eps_feas(), j);                            // I (x) Qeps  (mpc.py:531)
    }
}

// Bounds of row r exactly as mpc.py:551-580 / 404-408 build them, clipped to +-1e30 like osqp's wrapper.
// x0s: current x0; du0: bounds of the first nu Delta-u rows, [lower | upper] = Dumin/Dumax + u_{-1} (load_common).
__device__ __forceinline__ void row_bounds(const Ctx &c, const double *x0s, const double *du0, int r, double &lo, double &hi) {
    const Lay &L = c.L;
    if (r < L.rs) {
        lo = hi = (r < L.nx) ? -x0s[r] : 0.0;
    } else if (r < L.ri) {
        int j = r - L.rs; int k = idiv(j, L.rnx); int i = j - k * L.nx;
        lo = c.hot[L.oxmin + i]; hi = c.hot[L.oxmax + i];
    } else if (r < L.rdu) {
        int cc = r - L.ri; int k = idiv(cc, L.rnu); int jj = cc - k * L.nu;
        lo = c.hot[L.oumin + jj]; hi = c.hot[L.oumax + jj];
    } else {
        int rr = r - L.rdu; int k = idiv(rr, L.rnu); int jj = rr - k * L.nu;
        lo = c.hot[L.oDumin + jj]; hi = c.hot[L.oDumax + jj];
        if (rr < L.nu) { lo = du0[jj]; hi = du0[L.nu + jj]; }
    }
    lo = lo < -QP_INFTY ? -QP_INFTY : lo;
    hi = hi > QP_INFTY ? QP_INFTY : hi;
}

// Linear cost of the x and u variables (eps part is zero): mpc.py:489-526 / 411-452.
// um1s / xrs: LDS copies of u_{-1} and of a constant reference (load_common; xrs = nullptr: the reference is a trajectory, read from the step blob).
__device__ __forceinline__ void build_q(const Ctx &c, const double *step, double *Qv, const double *um1s = nullptr, const double *xrs = nullptr) {
    const Lay &L = c.L;
    const double *um1 = um1s ? um1s : step + L.nx, *xref = xrs ? xrs : step + L.nx + L.nu;
    const double *uref = c.hot + L.ouref;
    for (int j = threadIdx.x; j < L.n_x + L.n_u; j += NT) {
        double acc = 0.0;
        if (j < L.n_x) {
            int k = idiv(j, L.rnx), i = j - k * L.nx;
            const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
            if (L.xref_rows == 1) { for (int l = 0; l < L.nx; ++l) acc += Q[i * L.nx + l] * xref[l]; }
            else { const double *xr = xref + k * L.nx; for (int l = 0; l < L.nx; ++l) acc += xr[l] * Q[l * L.nx + i]; }
            acc = -acc;
        } else {
            int cc = j - L.n_x; int k = idiv(cc, L.rnu), jj = cc - k * L.nu;
            double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
            double a = 0.0;
            for (int l = 0; l < L.nu; ++l) a += c.Qu()[jj * L.nu + l] * uref[l];
            acc = iu * (-a);
            if (k == 0) { double d = 0.0; for (int l = 0; l < L.nu; ++l) d += c.QDu()[jj * L.nu + l] * um1[l]; acc += -d; }
        }
        Qv[j] = acc;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Block reductions (4 waves of 64).  red: LDS scratch of >= 4*K doubles.
// ------------------------------------------------------------------------------------------------
// Within a wave: four DPP steps (lane ^ 1, lane ^ 2, row rotations by 4 and 8) leave every lane with the result of its row of 16, the four rows
// are combined from their first lanes.  (__shfl_xor on a double is two ds_bpermute round trips per step: six dependent steps of those per value
// made the 12-value reduction of the termination check cost as much as one of its operator passes.)
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)xi, CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_mov_dpp((int)(xi >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)xi, l), hi = __builtin_amdgcn_readlane((int)(xi >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_reduce(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
    v = op(v, dpp_move_f64<0xB1>(v));            // quad_perm [1,0,3,2]
    v = op(v, dpp_move_f64<0x4E>(v));            // quad_perm [2,3,0,1]
    v = op(v, dpp_move_f64<0x124>(v));           // row_ror:4
    v = op(v, dpp_move_f64<0x128>(v));           // row_ror:8
    return op(op(readlane_f64(v, 0), readlane_f64(v, 16)), op(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// The same with the result in the LAST lane only: the four rows meet through the wave-level DPP broadcasts (lane 15 of a row into the next
// row: rows 1 and 3, then lane 31 into rows 2 and 3) instead of eight v_readlane and three more operations per value.  Lanes a broadcast does
// not write see 0 -- the identity here: sums, and maxima of non-negative numbers.  Pairs are combined as in wave_reduce ((r0, r1), (r2, r3),
// then the two): the same bits.
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_bcast_f64(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)xi, CTRL, ROWS, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(xi >> 32), CTRL, ROWS, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_reduce_last(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
    v = op(v, dpp_move_f64<0xB1>(v));            // quad_perm [1,0,3,2]
    v = op(v, dpp_move_f64<0x4E>(v));            // quad_perm [2,3,0,1]
    v = op(v, dpp_move_f64<0x124>(v));           // row_ror:4
    v = op(v, dpp_move_f64<0x128>(v));           // row_ror:8
    v = op(v, dpp_bcast_f64<0x142, 0xA>(v));     // row_bcast:15 -> rows 1, 3
    v = op(v, dpp_bcast_f64<0x143, 0xC>(v));     // row_bcast:31 -> rows 2, 3
    return v;
}
// The waves' partial results meet in LDS: thread i < K combines value i over the waves (in wave order: the sums are formed exactly as when
// every thread did this for all K values itself -- NWAVES x K dependent LDS reads per thread, a fifth of the termination check's reduction at
// four waves and more at eight) and publishes it; everybody reads K broadcast values.  red: >= (NWAVES + 1) K doubles.
template <int KMAX, int KSUM>
__device__ __forceinline__ void block_reduce(double *vmax, double *vsum, double *red) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int K = KMAX + KSUM;
    static_assert((NWAVES + 1) * K <= 16 * NWAVES, "Smem::red holds 16 doubles per wave");
#pragma unroll
    for (int i = 0; i < KMAX; ++i) vmax[i] = wave_reduce_last<true>(vmax[i]);
#pragma unroll
    for (int i = 0; i < KSUM; ++i) vsum[i] = wave_reduce_last<false>(vsum[i]);
    __syncthreads();
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) red[wv * K + i] = vmax[i];
#pragma unroll
        for (int i = 0; i < KSUM; ++i) red[wv * K + KMAX + i] = vsum[i];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int i = threadIdx.x;
        double v = red[i];
        if (i < KMAX) { for (int w = 1; w < NWAVES; ++w) v = fmax(v, red[w * K + i]); }
        else { for (int w = 1; w < NWAVES; ++w) v += red[w * K + i]; }
        red[NWAVES * K + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KMAX; ++i) vmax[i] = red[NWAVES * K + i];
#pragma unroll
    for (int i = 0; i < KSUM; ++i) vsum[i] = red[NWAVES * K + KMAX + i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Reduced KKT matrix  K = c P + diag(s) + A' diag(omega) A  with eps eliminated: stage blocks.
// Stage k holds v_k = (x_k, u_k) (u absent in the last stage); blocks are NB x NB, identity padded.
// ------------------------------------------------------------------------------------------------
// (DYN = false leaves out G' diag(om_dyn) G, G = [Ad Bd], om_dyn = the weights of the dynamics rows of stage k+1 -- the one term every
//  entry sums nx products for; mpcqp_wide.h forms it as a matrix product)
// (omd_copy: optional copy of om_dyn in LDS -- the one thing the nx-long sums would otherwise fetch from global memory entry by entry)
template <bool DYN = true>
__device__ __forceinline__ double kkt_diag_entry(const Ctx &c, const double *om, const double *sv, double cc, int k, int a, int b, const double *omd_copy = nullptr) {
    const Lay &L = c.L;
    const int nbk = (k < L.NcT) ? L.nb : L.nx;
    if (a >= nbk || b >= nbk) return a == b ? 1.0 : 0.0;
    const double *Ad = c.Ad(), *Bd = c.Bd();
    const double *omd = omd_copy ? omd_copy : om + (k + 1) * L.nx;          // dynamics rows of stage k+1
    double v = 0.0;
    if (a < L.nx && b < L.nx) {
        const double *Q = (k < L.Np) ? c.Qx() : c.QxN();
        v = cc * Q[min(a, b) * L.nx + max(a, b)];
        if (DYN && k < L.Np) for (int r = 0; r < L.nx; ++r) v += Ad[r * L.nx + a] * omd[r] * Ad[r * L.nx + b];
        if (a == b) {
            int e = k * L.nx + a;
            double ws = om[L.rs + e];
            if (L.soft) { const double ce = cc * c.eps_feas() + sv[L.oe + e]; ws *= ce / (ce + ws); }     // soft row with eps eliminated
            v += sv[e] + om[e] + ws;
        }
    } else if (a >= L.nx && b >= L.nx) {
        int ja = a - L.nx, jb = b - L.nx;
        double iu = (k == L.Nc - 1) ? (double)(L.Np - L.Nc + 1) : 1.0;
        double dk = (k == L.Nc - 1) ? 1.0 : 2.0;
        int lo = min(ja, jb), hi = max(ja, jb);
        v = cc * (iu * c.Qu()[lo * L.nu + hi] + dk * c.QDu()[lo * L.nu + hi]);
        if (DYN) for (int r = 0; r < L.nx; ++r) v += Bd[r * L.nu + ja] * omd[r] * Bd[r * L.nu + jb];
        int ca = k * L.nu + ja;
        const double *omdiff = om + L.rdu + L.nu;
        if (ja == jb) {
            v += sv[L.ou + ca] + om[L.ri + ca] + omdiff[ca];
            if (k == 0) v += om[L.rdu + ja];
            if (ca > 0) v += omdiff[ca - 1];
        } else if (hi - lo == 1) {
            v -= omdiff[k * L.nu + lo];
        }
    } else {
        int xa = a < L.nx ? a : b, ju = (a < L.nx ? b : a) - L.nx;
        if (DYN) for (int r = 0; r < L.nx; ++r) v += Ad[r * L.nx + xa] * omd[r] * Bd[r * L.nu + ju];
    }
    return v;
}

// K_{k+1,k}: rows = variables of stage k+1, cols = variables of stage k.
__device__ __forceinline__ double kkt_sub_entry(const Ctx &c, const double *om, double cc, int k, int a, int b) {
    const Lay &L = c.L;
    const int nbk = (k < L.NcT) ? L.nb : L.nx;
    const int nbn = (k + 1 < L.NcT) ? L.nb : L.nx;
    if (a >= nbn || b >= nbk) return 0.0;
    const double *omd = om + (k + 1) * L.nx;
    if (a < L.nx) {
        double co = (b < L.nx) ? c.Ad()[a * L.nx + b] : c.Bd()[a * L.nu + (b - L.nx)];
        return -omd[a] * co;
    }
    if (b < L.nx) return 0.0;
    int ja = a - L.nx, jb = b - L.nx;
    double v = -cc * c.QDu()[jb * L.nu + ja];                // mirror of the upper block -QDu
    if (ja == 0 && jb == L.nu - 1) v -= om[L.rdu + L.nu + k * L.nu + L.nu - 1];
    return v;
}

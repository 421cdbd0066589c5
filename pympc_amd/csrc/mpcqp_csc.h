// mpcqp_csc.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).  HOST code only.
// The solver seam with the caller's MATRICES (pyMPC/mpc.py:266: prob.setup(P, q, A, l, u, ...)): the controller data the kernels
// work from -- nx, nu, Np, Nc from the sparsity patterns; Ad, Bd, Qx, QxN, Qu, QDu, eps_feas from the values -- is read back out
// of P and A (the inverse of pyMPC/mpc.py:456-608), and the guarantee is a REBUILD: P and A are regenerated entry by entry from
// what was extracted and compared with what was given.  A QP that is not pyMPC's is refused (MPCQP_ERR_UNSUPPORTED); nothing
// is approximated.  (pympc_amd/qp_recover.py is the same logic in Python, kept for the CPU tests.)
#pragma once

struct CscPattern {
    int n = 0, m = 0;
    std::vector<int64_t> Pp, Ap;
    std::vector<int32_t> Pi, Ai;
};
// (one per handle created by mpcqp_create_csc, owned by the handle)
struct CscSeam { CscPattern pat; };

static int csc_dims(const CscPattern &c, int nx_hint, int nu_hint, int *nx, int *nu, int *Np, int *Nc, int *soft, std::string *why) {
    const int n = c.n, m = c.m;
    auto colcount = [&](int j) { return (int)(c.Ap[j + 1] - c.Ap[j]); };
    int n_x = 0;                                              // the slack columns: trailing columns with exactly one entry (their soft row)
    while (n_x < n && colcount(n - 1 - n_x) == 1) ++n_x;
    int n_u = n - 2 * n_x;
    *soft = 1;
    if (n_x == 0) {
        // SOFT_ON = False (mpc.py:237,530-597): no slack columns, n = n_x + n_u (an input column always has its box row and a Delta-u
        // row, so it never counts as one above).  The state box and the input box are then ONE identity block over all n columns,
        // rows n_x .. n_x + n - 1, and n_x is where it starts: the first row R >= 1 such that row R + i holds exactly the entry (R + i, i)
        // for every i (R = 0 -- the -1 of the x_0 rows -- stops at row nx, the first with entries of Ad and Bd).
        *soft = 0;
        std::vector<int> cnt(m, 0), col(m, -1);
        for (int j = 0; j < n; ++j) for (int64_t p = c.Ap[j]; p < c.Ap[j + 1]; ++p) { cnt[c.Ai[p]]++; col[c.Ai[p]] = j; }
        int R = 0;
        for (int r = 1; r + n <= m && !R; ++r) {
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) ok = cnt[r + i] == 1 && col[r + i] == i;
            if (ok) R = r;
        }
        n_x = R; n_u = n - n_x;
    }
    if (n_x < 2 || n_u < 1) { *why = "A has neither pyMPC's block of slack columns nor (SOFT_ON = False) its identity block of box rows"; return 1; }
    int u = nu_hint > 0 ? nu_hint : m - 2 * n_x - 2 * n_u;    // m = 2 n_x + n_u + (Nc + 1) nu
    if (u < 1 || n_u % u) { *why = "row/column counts do not fit m = 2 (Np+1) nx + Nc nu + (Nc+1) nu"; return 1; }
    int x = nx_hint;
    if (x <= 0) {         // rows 0..nx-1 of the dynamics block hold only the -1 of x_0 (and whatever explicit zeros scipy's kron left there,
                          // the same number in each); row nx is the first with more: the entries of Ad and Bd.  A wrong guess cannot
                          // survive: mpcqp_setup_csc rebuilds both matrices from what it reads with these dimensions.
        std::vector<int> per_row(n_x, 0);
        for (int j = 0; j < n; ++j) for (int64_t p = c.Ap[j]; p < c.Ap[j + 1]; ++p) if (c.Ai[p] < n_x) per_row[c.Ai[p]]++;
        x = 0;
        for (int r = 1; r < n_x; ++r) if (per_row[r] > per_row[0]) { x = r; break; }
    }
    if (x < 1 || n_x % x) { *why = "could not determine nx from the dynamics rows"; return 1; }
    *nx = x; *nu = u; *Np = n_x / x - 1; *Nc = n_u / u;
    if (*Np < 2 || *Nc < 1 || *Nc > *Np || m != 2 * n_x + n_u + (*Nc + 1) * u) { *why = "shapes do not fit an MPC QP"; return 1; }
    return 0;
}

// value stored at (r, c) of a CSC matrix (0 if absent)
static double csc_at(const std::vector<int64_t> &cp, const std::vector<int32_t> &ri, const double *val, int r, int c) {
    for (int64_t p = cp[c]; p < cp[c + 1]; ++p) if (ri[p] == r) return val[p];
    return 0.0;
}

struct MpcBlocks { int nx, nu, Np, Nc, soft; std::vector<double> Ad, Bd, Qx, QxN, Qu, QDu; double eps_feas; };

// what pyMPC's builder puts at (r, c) of A / of the upper triangle of P, from the blocks (the host twin of A_row / P_row in mpcqp_qp.h)
static double A_expected(const MpcBlocks &b, int r, int c) {
    const int nx = b.nx, nu = b.nu, n_x = (b.Np + 1) * nx, n_u = b.Nc * nu, ou = n_x, oe = n_x + n_u, rs = n_x, ri = 2 * n_x, rdu = 2 * n_x + n_u;
    if (r < rs) {
        const int k = r / nx, i = r % nx;
        if (c == r) return -1.0;
        if (k > 0) {
            if (c >= (k - 1) * nx && c < k * nx) return b.Ad[i * nx + (c - (k - 1) * nx)];
            const int ku = std::min(k - 1, b.Nc - 1), base = ou + ku * nu;
            if (c >= base && c < base + nu) return b.Bd[i * nu + (c - base)];
        }
        return 0.0;
    }
    if (r < ri) { const int j = r - rs; return (c == j || (b.soft && c == oe + j)) ? 1.0 : 0.0; }
    if (r < rdu) return c == ou + (r - ri) ? 1.0 : 0.0;
    const int rr = r - rdu;
    if (rr < nu) return c == ou + rr ? 1.0 : 0.0;
    const int cc = rr - nu;
    if (c == ou + cc) return -1.0;
    if (cc + 1 < n_u && c == ou + cc + 1) return 1.0;
    return 0.0;
}
static double P_expected_upper(const MpcBlocks &b, int r, int c) {      // r <= c
    const int nx = b.nx, nu = b.nu, n_x = (b.Np + 1) * nx, n_u = b.Nc * nu, ou = n_x, oe = n_x + n_u;
    if (r < ou) {
        const int k = r / nx, i = r % nx;
        if (c >= k * nx && c < (k + 1) * nx) return (k < b.Np ? b.Qx : b.QxN)[i * nx + (c - k * nx)];
        return 0.0;
    }
    if (r < oe) {
        const int cc = r - ou, k = cc / nu, jj = cc % nu, base = ou + k * nu;
        if (c >= base && c < base + nu) {
            const double iu = (k == b.Nc - 1) ? (double)(b.Np - b.Nc + 1) : 1.0, dk = (k == b.Nc - 1) ? 1.0 : 2.0;
            const int l = c - base;
            return iu * b.Qu[jj * nu + l] + dk * b.QDu[jj * nu + l];
        }
        if (k + 1 < b.Nc && c >= base + nu && c < base + 2 * nu) return -b.QDu[jj * nu + (c - base - nu)];
        return 0.0;
    }
    return c == r ? b.eps_feas : 0.0;
}

// blocks of one instance out of its values, verified by the rebuild; l, u checked for the stage-periodic structure of mpc.py:551-580
static int csc_recover(const CscPattern &c, int nx, int nu, int Np, int Nc, int soft, const double *Pv, const double *Av, const double *q,
                       const double *l, const double *u, MpcBlocks *out, std::string *why) {
    MpcBlocks b; b.nx = nx; b.nu = nu; b.Np = Np; b.Nc = Nc; b.soft = soft;
    const int n_x = (Np + 1) * nx, n_u = Nc * nu, n = c.n, m = c.m;
    auto Pf = [&](int r, int cc) { return r <= cc ? csc_at(c.Pp, c.Pi, Pv, r, cc) : csc_at(c.Pp, c.Pi, Pv, cc, r); };      // what a solver keeping triu(P) sees
    auto Aat = [&](int r, int cc) { return csc_at(c.Ap, c.Ai, Av, r, cc); };
    b.Ad.resize(nx * nx); b.Bd.resize(nx * nu); b.Qx.resize(nx * nx); b.QxN.resize(nx * nx); b.Qu.resize(nu * nu); b.QDu.resize(nu * nu);
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) { b.Ad[i * nx + j] = Aat(nx + i, j); b.Qx[i * nx + j] = Pf(i, j); b.QxN[i * nx + j] = Pf(Np * nx + i, Np * nx + j); }
        for (int j = 0; j < nu; ++j) b.Bd[i * nu + j] = Aat(nx + i, n_x + j);
    }
    double d0max = 0.0;
    for (int i = 0; i < nu; ++i) for (int j = 0; j < nu; ++j) {
        const double D0 = Pf(n_x + i, n_x + j);
        d0max = std::max(d0max, fabs(D0));
        if (Nc >= 2) { b.QDu[i * nu + j] = -Pf(n_x + i, n_x + nu + j); b.Qu[i * nu + j] = D0 - 2.0 * b.QDu[i * nu + j]; }
        else { b.QDu[i * nu + j] = 0.0; b.Qu[i * nu + j] = D0 / (double)Np; }      // one block iU Qu + QDu: any split gives the same P (q is the caller's)
    }
    b.eps_feas = soft ? Pf(n_x + n_u, n_x + n_u) : 1e6;      // (no slack block without soft constraints: the value is never used)
    // ---- the guarantee: rebuild and compare, both ways (stored nonzero -> expected value; expected nonzero -> stored)
    int badA = 0, badP = 0;
    for (int j = 0; j < n; ++j) for (int64_t p = c.Ap[j]; p < c.Ap[j + 1]; ++p) if (Av[p] != A_expected(b, c.Ai[p], j)) ++badA;
    for (int r = 0; r < m && !badA; ++r) {
        // the expected nonzeros of row r live in a handful of columns: probe them
        const int rs = n_x, ri = 2 * n_x, rdu = 2 * n_x + n_u;
        auto probe = [&](int cc) { if (cc >= 0 && cc < n) { const double e = A_expected(b, r, cc); if (e != 0.0 && Aat(r, cc) != e) ++badA; } };
        if (r < rs) { const int k = r / nx; probe(r); if (k > 0) { for (int t = 0; t < nx; ++t) probe((k - 1) * nx + t); for (int t = 0; t < nu; ++t) probe(n_x + std::min(k - 1, Nc - 1) * nu + t); } }
        else if (r < ri) { probe(r - rs); if (soft) probe(n_x + n_u + r - rs); }
        else if (r < rdu) probe(n_x + r - ri);
        else { const int rr = r - rdu; if (rr < nu) probe(n_x + rr); else { probe(n_x + rr - nu); probe(n_x + rr - nu + 1); } }
    }
    // The input-weight blocks are stored as sums (Qu + 2 QDu, mpc.py:505-526): Qu = D0 - 2 QDu carries D0's rounding error, and re-adding
    // may move the last bit.  With Nc < Np the LAST input block is rebuilt as (Np - Nc + 1) Qu + QDu (mpc.py:513-517), which multiplies that
    // error by Np - Nc + 1: the bound for that block is scaled by the same factor.
    const double ptol = 4.0 * 2.220446049250313e-16 * std::max(1.0, d0max);
    const double ptol_last = ptol * (double)(Np - Nc + 1);
    auto p_ok = [&](int r, int cc, double given) {
        const double e = P_expected_upper(b, r, cc);
        if (given == e) return true;
        if (!(r >= n_x && r < n_x + n_u && cc >= n_x && cc < n_x + n_u)) return false;
        const bool last = r >= n_x + (Nc - 1) * nu && cc >= n_x + (Nc - 1) * nu;      // the diagonal block of the held input
        return fabs(given - e) <= (last ? ptol_last : ptol);
    };
    for (int j = 0; j < n; ++j) for (int64_t p = c.Pp[j]; p < c.Pp[j + 1]; ++p) if (c.Pi[p] <= j && !p_ok(c.Pi[p], j, Pv[p])) ++badP;
    for (int r = 0; r < n && !badP; ++r) {
        const int lo = r, hi = r < n_x ? (r / nx + 1) * nx : (r < n_x + n_u ? std::min(n_x + n_u, n_x + ((r - n_x) / nu + 2) * nu) : r + 1);
        for (int cc = lo; cc < hi; ++cc) { const double e = P_expected_upper(b, r, cc); if (e != 0.0 && !p_ok(r, cc, csc_at(c.Pp, c.Pi, Pv, r, cc))) ++badP; }
    }
    if (badA || badP) { *why = "P, A are not the matrices pyMPC builds from their own blocks (" + std::to_string(badP) + " / " + std::to_string(badA) + " entries differ)"; return 1; }
    // ---- vectors: zero slack part of q; l, u stage-periodic, equality rows l == u, zero behind the first nx
    const int rs = n_x, ri = 2 * n_x, rdu = 2 * n_x + n_u;
    for (int j = n_x + n_u; j < n; ++j) if (q[j] != 0.0) { *why = "q must have a zero slack part (mpc.py:599)"; return 1; }
    for (int pass = 0; pass < 2; ++pass) {
        const double *v = pass ? u : l;
        bool ok = true;
        for (int r = nx; r < n_x; ++r) ok &= v[r] == 0.0;
        for (int r = rs; r < ri; ++r) ok &= v[r] == v[rs + (r - rs) % nx];
        for (int r = ri; r < rdu; ++r) ok &= v[r] == v[ri + (r - ri) % nu];
        for (int r = rdu + nu; r < m; ++r) ok &= v[r] == v[rdu + nu + (r - rdu - nu) % nu];
        if (!ok) { *why = std::string(pass ? "u" : "l") + " does not have the stage-periodic structure of pyMPC/mpc.py:551-580"; return 1; }
    }
    for (int r = 0; r < nx; ++r) if (l[r] != u[r]) { *why = "the initial-state rows must be equalities (l[:nx] == u[:nx] = -x0)"; return 1; }
    *out = std::move(b);
    return 0;
}

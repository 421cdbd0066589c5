/*
 * osqp_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-thread, FP64 restatement of the solver the reference calls at
 * pyMPC/mpc.py:241,266,369,454: the third-party package `osqp` (un-vendored and UNPINNED in the
 * reference, setup.py:11; API era 0.5/0.6 by its keyword set).  `osqp` is absent from
 * /root/reference and from this image, so this file restates its PUBLISHED algorithm
 * (Stellato, Banjac, Goulart, Bemporad, Boyd: "OSQP: an operator splitting solver for quadratic
 * programs", Math. Prog. Comp. 12, 2020: Algorithm 1, sections 3.4 (termination), 5.1 (Ruiz
 * equilibration), 5.2 (rho selection/adaptation), 3.5 (infeasibility certificates)) with the
 * documented 0.6.x defaults:  rho=0.1, sigma=1e-6, alpha=1.6, max_iter=4000, eps_abs=eps_rel=1e-3,
 * eps_prim_inf=eps_dual_inf=1e-4, scaling=10, adaptive_rho=1 (tolerance 5), check_termination=25,
 * scaled_termination=0, warm_start=1, polish=0; equality rows get 1e3*rho, free rows rho_min=1e-6;
 * bounds beyond +-1e30 are infinite.  The linear system is the quasi-definite KKT matrix
 * [P+sigma I, A'; A, -diag(rho)^-1] factored by a sparse up-looking LDL' (elimination tree +
 * row-by-row numeric phase, the classical algorithm of Davis' "Algorithm 849: a concise sparse
 * Cholesky factorization package") under a caller-supplied fill-reducing permutation.
 *
 * PARITY STATUS: "parity unpinned" against a real OSQP binary (none reachable; tests/test_real_osqp.py holds the
 * comparison and skips until `import osqp` works).  What pins this oracle instead: (1) it consumes P,q,A,l,u that are
 * bit-identical to what the reference builds (tests/golden/qp_*.npz, captured from the imported reference); (2) every
 * optimum it returns at tight tolerance is certified solver-independently by KKT conditions and cross-checked with
 * the HiGHS QP solver bundled in scipy (tests/test_oracle.py, tests/golden/make_optimum.py); (3) where no inequality
 * is active it reproduces the condensed closed-form controller of test_scripts/alternative/unconstrained.py:141-183,
 * restated with dense linear algebra only (tests/closed_form.py, tests/test_closed_form.py); (4) plugged into the
 * REFERENCE's own controller class as its `osqp` -- and next to the reference's own LinearStateEstimator -- it produces
 * the closed-loop golden trajectories of tests/golden/make_traj.py (traj_*.npz), which the product is tested against.
 * `adaptive_rho_interval=0` (OSQP: derived from wall-clock setup time, hence not reproducible)
 * is resolved deterministically to 4*check_termination, OSQP's own rule for builds without timers.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime (closed-loop driver at the end of the file) */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4

enum {
    ST_UNSOLVED = -10, ST_SOLVED = 1, ST_SOLVED_INACCURATE = 2, ST_PRIMAL_INFEASIBLE = -3,
    ST_PRIMAL_INFEASIBLE_INACCURATE = 3, ST_DUAL_INFEASIBLE = -4, ST_DUAL_INFEASIBLE_INACCURATE = 4,
    ST_MAX_ITER_REACHED = -2, ST_NON_CVX = -7
};

typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, adaptive_rho_tolerance;
    int max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval, warm_start,
        scaled_termination;
} oracle_settings;

typedef struct {
    int status, iter, rho_updates;
    double obj_val, pri_res, dua_res, rho_estimate;
} oracle_info;

typedef struct {
    int64_t n, m;
    int64_t *Pp, *Pi; double *Px;           /* upper triangle, CSC, SCALED in place */
    int64_t *Ap, *Ai; double *Ax;           /* CSC, SCALED in place */
    double *q, *l, *u;                      /* scaled */
    double *D, *E, *Dinv, *Einv, c, cinv;
    double *rho_vec, *rho_inv_vec; int *constr_type;
    oracle_settings s;
    /* iterates (scaled) */
    double *x, *z, *y, *x_prev, *z_prev, *xz_tilde, *delta_x, *delta_y;
    double *Ax_, *Px_, *Aty, *Atdy, *Pdx, *Adx, *D_temp, *D_temp_A, *E_temp;
    /* KKT + factor */
    int64_t nk, *perm, *iperm;              /* perm[new] = old */
    int64_t *Kp, *Ki; double *Kx;           /* permuted upper-tri CSC */
    int64_t *rho_pos;                       /* position in Kx of the (n+i, n+i) diagonal entry */
    int64_t *Lp, *Li, *Parent, *Lnz, *Flag, *Pattern; double *Lx, *Dk, *Dkinv, *Y, *rhs, *sol;
    int64_t nnzL;
    oracle_info info;
    int first_run;
} oracle_work;

/* ------------------------------------------------------------------ small dense helpers */
static double norm_inf(const double *v, int64_t n) {
    double r = 0; for (int64_t i = 0; i < n; i++) { double a = fabs(v[i]); if (a > r) r = a; } return r;
}
static double scaled_norm_inf(const double *s, const double *v, int64_t n) {
    double r = 0; for (int64_t i = 0; i < n; i++) { double a = fabs(s[i] * v[i]); if (a > r) r = a; } return r;
}
static double dot(const double *a, const double *b, int64_t n) {
    double r = 0; for (int64_t i = 0; i < n; i++) r += a[i] * b[i]; return r;
}
/* y = A x (CSC) */
static void mat_vec(int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                    const double *x, double *y) {
    for (int64_t i = 0; i < m; i++) y[i] = 0;
    for (int64_t j = 0; j < n; j++) for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) y[Ai[p]] += Ax[p] * x[j];
}
/* y = A' x */
static void mat_tpose_vec(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                          const double *x, double *y) {
    for (int64_t j = 0; j < n; j++) { double s = 0; for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * x[Ai[p]]; y[j] = s; }
}
/* y = P x with P symmetric given by its upper triangle */
static void sym_mat_vec(int64_t n, const int64_t *Pp, const int64_t *Pi, const double *Px,
                        const double *x, double *y) {
    for (int64_t i = 0; i < n; i++) y[i] = 0;
    for (int64_t j = 0; j < n; j++) for (int64_t p = Pp[j]; p < Pp[j + 1]; p++) {
        int64_t i = Pi[p];
        y[i] += Px[p] * x[j];
        if (i != j) y[j] += Px[p] * x[i];
    }
}
static double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

/* ------------------------------------------------------------------ Ruiz equilibration (paper 5.1) */
static void scale_data(oracle_work *w) {
    int64_t n = w->n, m = w->m;
    for (int64_t i = 0; i < n; i++) w->D[i] = 1.0;
    for (int64_t i = 0; i < m; i++) w->E[i] = 1.0;
    w->c = 1.0;
    for (int it = 0; it < w->s.scaling; it++) {
        double *Dt = w->D_temp, *DtA = w->D_temp_A, *Et = w->E_temp;
        /* inf-norm of the columns of the KKT matrix [P A'; A 0] */
        for (int64_t j = 0; j < n; j++) { Dt[j] = 0; DtA[j] = 0; }
        for (int64_t i = 0; i < m; i++) Et[i] = 0;
        for (int64_t j = 0; j < n; j++) for (int64_t p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
            int64_t i = w->Pi[p]; double a = fabs(w->Px[p]);
            if (a > Dt[j]) Dt[j] = a;
            if (i != j && a > Dt[i]) Dt[i] = a;
        }
        for (int64_t j = 0; j < n; j++) for (int64_t p = w->Ap[j]; p < w->Ap[j + 1]; p++) {
            double a = fabs(w->Ax[p]);
            if (a > DtA[j]) DtA[j] = a;
            if (a > Et[w->Ai[p]]) Et[w->Ai[p]] = a;
        }
        for (int64_t j = 0; j < n; j++) { double v = Dt[j] > DtA[j] ? Dt[j] : DtA[j]; Dt[j] = 1.0 / sqrt(limit_scaling(v)); }
        for (int64_t i = 0; i < m; i++) Et[i] = 1.0 / sqrt(limit_scaling(Et[i]));
        /* P <- Dt P Dt, A <- Et A Dt, q <- Dt q */
        for (int64_t j = 0; j < n; j++) for (int64_t p = w->Pp[j]; p < w->Pp[j + 1]; p++) w->Px[p] *= Dt[w->Pi[p]] * Dt[j];
        for (int64_t j = 0; j < n; j++) for (int64_t p = w->Ap[j]; p < w->Ap[j + 1]; p++) w->Ax[p] *= Et[w->Ai[p]] * Dt[j];
        for (int64_t j = 0; j < n; j++) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
        for (int64_t i = 0; i < m; i++) w->E[i] *= Et[i];
        /* cost scaling: 1 / max(mean column norm of P, ||q||_inf) */
        for (int64_t j = 0; j < n; j++) Dt[j] = 0;
        for (int64_t j = 0; j < n; j++) for (int64_t p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
            int64_t i = w->Pi[p]; double a = fabs(w->Px[p]);
            if (a > Dt[j]) Dt[j] = a;
            if (i != j && a > Dt[i]) Dt[i] = a;
        }
        double c_temp = 0; for (int64_t j = 0; j < n; j++) c_temp += Dt[j];
        c_temp /= (double)n;
        double qn = limit_scaling(norm_inf(w->q, n));
        c_temp = limit_scaling(c_temp > qn ? c_temp : qn);
        c_temp = 1.0 / c_temp;
        for (int64_t p = 0; p < w->Pp[n]; p++) w->Px[p] *= c_temp;
        for (int64_t j = 0; j < n; j++) w->q[j] *= c_temp;
        w->c *= c_temp;
    }
    for (int64_t j = 0; j < n; j++) w->Dinv[j] = 1.0 / w->D[j];
    for (int64_t i = 0; i < m; i++) w->Einv[i] = 1.0 / w->E[i];
    w->cinv = 1.0 / w->c;
    for (int64_t i = 0; i < m; i++) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
}

/* ------------------------------------------------------------------ rho vector (paper 5.2) */
static int set_rho_vec(oracle_work *w, int only_if_changed) {
    int changed = 0;
    for (int64_t i = 0; i < w->m; i++) {
        int t; double r;
        if (w->l[i] < -QP_INFTY * MIN_SCALING && w->u[i] > QP_INFTY * MIN_SCALING) { t = -1; r = RHO_MIN; }
        else if (w->u[i] - w->l[i] < RHO_TOL) { t = 1; r = RHO_EQ_OVER_RHO_INEQ * w->s.rho; }
        else { t = 0; r = w->s.rho; }
        if (only_if_changed) { if (t != w->constr_type[i]) changed = 1; else continue; }
        w->constr_type[i] = t; w->rho_vec[i] = r; w->rho_inv_vec[i] = 1.0 / r;
    }
    return changed;
}

/* ------------------------------------------------------------------ sparse LDL' */
static void ldl_symbolic(oracle_work *w) {
    int64_t n = w->nk;
    for (int64_t k = 0; k < n; k++) {
        w->Parent[k] = -1; w->Flag[k] = k; w->Lnz[k] = 0;
        for (int64_t p = w->Kp[k]; p < w->Kp[k + 1]; p++) {
            int64_t i = w->Ki[p];
            if (i < k) for (; w->Flag[i] != k; i = w->Parent[i]) {
                if (w->Parent[i] == -1) w->Parent[i] = k;
                w->Lnz[i]++; w->Flag[i] = k;
            }
        }
    }
    w->Lp[0] = 0;
    for (int64_t k = 0; k < n; k++) w->Lp[k + 1] = w->Lp[k] + w->Lnz[k];
    w->nnzL = w->Lp[n];
}
static int ldl_numeric(oracle_work *w) {
    int64_t n = w->nk;
    double *Y = w->Y;
    for (int64_t k = 0; k < n; k++) {
        Y[k] = 0; int64_t top = n; w->Flag[k] = k; w->Lnz[k] = 0;
        for (int64_t p = w->Kp[k]; p < w->Kp[k + 1]; p++) {
            int64_t i = w->Ki[p];
            if (i <= k) {
                Y[i] += w->Kx[p];
                int64_t len = 0;
                for (; w->Flag[i] != k; i = w->Parent[i]) { w->Pattern[len++] = i; w->Flag[i] = k; }
                while (len > 0) w->Pattern[--top] = w->Pattern[--len];
            }
        }
        w->Dk[k] = Y[k]; Y[k] = 0;
        for (; top < n; top++) {
            int64_t i = w->Pattern[top]; double yi = Y[i]; Y[i] = 0;
            int64_t p2 = w->Lp[i] + w->Lnz[i];
            for (int64_t p = w->Lp[i]; p < p2; p++) Y[w->Li[p]] -= w->Lx[p] * yi;
            double lki = yi / w->Dk[i];
            w->Dk[k] -= lki * yi;
            w->Li[p2] = k; w->Lx[p2] = lki; w->Lnz[i]++;
        }
        if (w->Dk[k] == 0.0) return -1;
        w->Dkinv[k] = 1.0 / w->Dk[k];
    }
    return 0;
}
static void ldl_solve(oracle_work *w, double *b) {      /* in place, b in permuted order */
    int64_t n = w->nk;
    for (int64_t j = 0; j < n; j++) { double bj = b[j]; for (int64_t p = w->Lp[j]; p < w->Lp[j + 1]; p++) b[w->Li[p]] -= w->Lx[p] * bj; }
    for (int64_t j = 0; j < n; j++) b[j] *= w->Dkinv[j];
    for (int64_t j = n - 1; j >= 0; j--) { double s = b[j]; for (int64_t p = w->Lp[j]; p < w->Lp[j + 1]; p++) s -= w->Lx[p] * b[w->Li[p]]; b[j] = s; }
}

/* Build the permuted upper-triangular KKT matrix from the scaled data. */
typedef struct { int64_t r, c; double v; int64_t tag; } trip;
static int trip_cmp(const void *a, const void *b) {
    const trip *x = (const trip *)a, *y = (const trip *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return 0;
}
static int build_kkt(oracle_work *w) {
    int64_t n = w->n, m = w->m, nk = n + m;
    int64_t cap = w->Pp[n] + n + w->Ap[n] + m, nt = 0;
    trip *T = (trip *)malloc(sizeof(trip) * (size_t)cap);
    if (!T) return -1;
    /* P + sigma I (make sure every diagonal entry exists) */
    for (int64_t j = 0; j < n; j++) {
        for (int64_t p = w->Pp[j]; p < w->Pp[j + 1]; p++) { T[nt].r = w->Pi[p]; T[nt].c = j; T[nt].v = w->Px[p]; T[nt].tag = -1; nt++; }
        T[nt].r = j; T[nt].c = j; T[nt].v = w->s.sigma; T[nt].tag = -1; nt++;
    }
    for (int64_t j = 0; j < n; j++) for (int64_t p = w->Ap[j]; p < w->Ap[j + 1]; p++) {
        T[nt].r = j; T[nt].c = n + w->Ai[p]; T[nt].v = w->Ax[p]; T[nt].tag = -1; nt++;     /* A' block (upper) */
    }
    for (int64_t i = 0; i < m; i++) { T[nt].r = n + i; T[nt].c = n + i; T[nt].v = -w->rho_inv_vec[i]; T[nt].tag = i; nt++; }
    /* permute, keep upper triangle */
    for (int64_t t = 0; t < nt; t++) {
        int64_t r = w->iperm[T[t].r], c = w->iperm[T[t].c];
        if (r > c) { int64_t tmp = r; r = c; c = tmp; }
        T[t].r = r; T[t].c = c;
    }
    qsort(T, (size_t)nt, sizeof(trip), trip_cmp);
    w->Kp = (int64_t *)calloc((size_t)nk + 1, sizeof(int64_t));
    w->Ki = (int64_t *)malloc(sizeof(int64_t) * (size_t)nt);
    w->Kx = (double *)malloc(sizeof(double) * (size_t)nt);
    int64_t nz = 0;
    for (int64_t t = 0; t < nt; t++) {
        if (nz > 0 && t > 0 && T[t].c == T[t - 1].c && T[t].r == T[t - 1].r) { w->Kx[nz - 1] += T[t].v; }
        else { w->Ki[nz] = T[t].r; w->Kx[nz] = T[t].v; w->Kp[T[t].c + 1]++; nz++; }
        if (T[t].tag >= 0) w->rho_pos[T[t].tag] = nz - 1;
    }
    for (int64_t k = 0; k < nk; k++) w->Kp[k + 1] += w->Kp[k];
    free(T);
    return 0;
}
static void update_kkt_rho(oracle_work *w) { for (int64_t i = 0; i < w->m; i++) w->Kx[w->rho_pos[i]] = -w->rho_inv_vec[i]; }

/* ------------------------------------------------------------------ public API */
void oracle_default_settings(oracle_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
    s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->adaptive_rho_tolerance = 5.0;
    s->max_iter = 4000; s->check_termination = 25; s->scaling = 10; s->adaptive_rho = 1;
    s->adaptive_rho_interval = 0; s->warm_start = 1; s->scaled_termination = 0;
}

static double *dvec(int64_t n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static int64_t *ivec(int64_t n) { return (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t)); }

void oracle_free(oracle_work *w);

/* P: upper triangle CSC. perm: fill-reducing ordering of the (n+m) KKT matrix, perm[new]=old, or NULL. */
oracle_work *oracle_setup(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px,
                          const int64_t *Ap, const int64_t *Ai, const double *Ax,
                          const double *q, const double *l, const double *u,
                          const int64_t *perm, const oracle_settings *s) {
    oracle_work *w = (oracle_work *)calloc(1, sizeof(oracle_work));
    w->n = n; w->m = m; w->s = *s; w->nk = n + m;
    if (w->s.adaptive_rho && w->s.adaptive_rho_interval == 0)
        w->s.adaptive_rho_interval = w->s.check_termination ? 4 * w->s.check_termination : 100;
    int64_t nnzP = Pp[n], nnzA = Ap[n];
    w->Pp = ivec(n + 1); w->Pi = ivec(nnzP); w->Px = dvec(nnzP);
    w->Ap = ivec(n + 1); w->Ai = ivec(nnzA); w->Ax = dvec(nnzA);
    memcpy(w->Pp, Pp, sizeof(int64_t) * (size_t)(n + 1)); memcpy(w->Pi, Pi, sizeof(int64_t) * (size_t)nnzP); memcpy(w->Px, Px, sizeof(double) * (size_t)nnzP);
    memcpy(w->Ap, Ap, sizeof(int64_t) * (size_t)(n + 1)); memcpy(w->Ai, Ai, sizeof(int64_t) * (size_t)nnzA); memcpy(w->Ax, Ax, sizeof(double) * (size_t)nnzA);
    w->q = dvec(n); w->l = dvec(m); w->u = dvec(m);
    memcpy(w->q, q, sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < m; i++) {      /* the Python wrapper of OSQP clips infinite bounds */
        w->l[i] = l[i] < -QP_INFTY ? -QP_INFTY : l[i];
        w->u[i] = u[i] > QP_INFTY ? QP_INFTY : u[i];
    }
    w->D = dvec(n); w->Dinv = dvec(n); w->E = dvec(m); w->Einv = dvec(m);
    w->D_temp = dvec(n); w->D_temp_A = dvec(n); w->E_temp = dvec(m);
    w->rho_vec = dvec(m); w->rho_inv_vec = dvec(m); w->constr_type = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    w->x = dvec(n); w->z = dvec(m); w->y = dvec(m); w->x_prev = dvec(n); w->z_prev = dvec(m);
    w->xz_tilde = dvec(n + m); w->delta_x = dvec(n); w->delta_y = dvec(m);
    w->Ax_ = dvec(m); w->Px_ = dvec(n); w->Aty = dvec(n); w->Atdy = dvec(n); w->Pdx = dvec(n); w->Adx = dvec(m);
    if (w->s.scaling) scale_data(w);
    else {
        for (int64_t j = 0; j < n; j++) w->D[j] = w->Dinv[j] = 1.0;
        for (int64_t i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.0;
        w->c = w->cinv = 1.0;
    }
    set_rho_vec(w, 0);
    /* permutation */
    w->perm = ivec(w->nk); w->iperm = ivec(w->nk);
    for (int64_t k = 0; k < w->nk; k++) w->perm[k] = perm ? perm[k] : k;
    for (int64_t k = 0; k < w->nk; k++) w->iperm[w->perm[k]] = k;
    w->rho_pos = ivec(m);
    if (build_kkt(w)) { oracle_free(w); return NULL; }
    w->Lp = ivec(w->nk + 1); w->Parent = ivec(w->nk); w->Lnz = ivec(w->nk); w->Flag = ivec(w->nk); w->Pattern = ivec(w->nk);
    w->Dk = dvec(w->nk); w->Dkinv = dvec(w->nk); w->Y = dvec(w->nk); w->rhs = dvec(w->nk); w->sol = dvec(w->nk);
    ldl_symbolic(w);
    w->Li = ivec(w->nnzL); w->Lx = dvec(w->nnzL);
    if (ldl_numeric(w)) { oracle_free(w); return NULL; }
    w->info.status = ST_UNSOLVED; w->info.rho_estimate = w->s.rho;
    w->first_run = 1;
    return w;
}

void oracle_free(oracle_work *w) {
    if (!w) return;
    void *ptrs[] = { w->Pp, w->Pi, w->Px, w->Ap, w->Ai, w->Ax, w->q, w->l, w->u, w->D, w->E, w->Dinv, w->Einv,
        w->rho_vec, w->rho_inv_vec, w->constr_type, w->x, w->z, w->y, w->x_prev, w->z_prev, w->xz_tilde,
        w->delta_x, w->delta_y, w->Ax_, w->Px_, w->Aty, w->Atdy, w->Pdx, w->Adx, w->D_temp, w->D_temp_A, w->E_temp,
        w->perm, w->iperm, w->Kp, w->Ki, w->Kx, w->rho_pos, w->Lp, w->Li, w->Parent, w->Lnz, w->Flag, w->Pattern,
        w->Lx, w->Dk, w->Dkinv, w->Y, w->rhs, w->sol };
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
    free(w);
}

int64_t oracle_nnzL(const oracle_work *w) { return w->nnzL; }

/* osqp.update(q=, l=, u=): any pointer may be NULL.  Returns 0, or 1 if l > u somewhere. */
int oracle_update(oracle_work *w, const double *q, const double *l, const double *u) {
    int64_t n = w->n, m = w->m;
    if (q) { for (int64_t j = 0; j < n; j++) w->q[j] = w->c * w->D[j] * q[j]; w->info.status = ST_UNSOLVED; }
    if (l || u) {
        for (int64_t i = 0; i < m; i++) {
            double li = l ? (l[i] < -QP_INFTY ? -QP_INFTY : l[i]) : w->l[i] * w->Einv[i];
            double ui = u ? (u[i] > QP_INFTY ? QP_INFTY : u[i]) : w->u[i] * w->Einv[i];
            if (li > ui) return 1;
            w->l[i] = w->E[i] * li; w->u[i] = w->E[i] * ui;
        }
        w->info.status = ST_UNSOLVED;
        if (set_rho_vec(w, 1)) { update_kkt_rho(w); ldl_numeric(w); }
    }
    return 0;
}

/* osqp.update_settings(): tolerances, iteration limits, alpha, the adaptive-rho rule -- what may change after setup
 * (rho, sigma and the scaling shaped the factorization and stay). */
void oracle_set_tolerances(oracle_work *w, const oracle_settings *s) {
    const double rho = w->s.rho, sigma = w->s.sigma; const int scaling = w->s.scaling;
    w->s = *s; w->s.rho = rho; w->s.sigma = sigma; w->s.scaling = scaling;
    if (w->s.adaptive_rho && w->s.adaptive_rho_interval == 0)
        w->s.adaptive_rho_interval = w->s.check_termination ? 4 * w->s.check_termination : 100;
}

void oracle_warm_start(oracle_work *w, const double *x, const double *y) {
    int64_t n = w->n, m = w->m;
    if (x) { for (int64_t j = 0; j < n; j++) w->x[j] = w->Dinv[j] * x[j]; mat_vec(m, n, w->Ap, w->Ai, w->Ax, w->x, w->z); }
    if (y) for (int64_t i = 0; i < m; i++) w->y[i] = w->c * w->Einv[i] * y[i];
}

void oracle_get_scaling(const oracle_work *w, double *D, double *E, double *c) {
    memcpy(D, w->D, sizeof(double) * (size_t)w->n); memcpy(E, w->E, sizeof(double) * (size_t)w->m); *c = w->c;
}
/* current scaled iterate and rho (for iterate-level comparisons in the tests) */
void oracle_get_iterate(const oracle_work *w, double *x, double *z, double *y, double *rho) {
    memcpy(x, w->x, sizeof(double) * (size_t)w->n); memcpy(z, w->z, sizeof(double) * (size_t)w->m);
    memcpy(y, w->y, sizeof(double) * (size_t)w->m); *rho = w->s.rho;
}

static void update_info(oracle_work *w) {
    int64_t n = w->n, m = w->m;
    /* objective */
    sym_mat_vec(n, w->Pp, w->Pi, w->Px, w->x, w->Px_);
    double obj = 0.5 * dot(w->x, w->Px_, n) + dot(w->q, w->x, n);
    w->info.obj_val = w->s.scaling ? obj * w->cinv : obj;
    /* primal residual (z_prev is the work vector) */
    mat_vec(m, n, w->Ap, w->Ai, w->Ax, w->x, w->Ax_);
    for (int64_t i = 0; i < m; i++) w->z_prev[i] = w->Ax_[i] - w->z[i];
    w->info.pri_res = (w->s.scaling && !w->s.scaled_termination) ? scaled_norm_inf(w->Einv, w->z_prev, m) : norm_inf(w->z_prev, m);
    /* dual residual (x_prev is the work vector) */
    mat_tpose_vec(n, w->Ap, w->Ai, w->Ax, w->y, w->Aty);
    for (int64_t j = 0; j < n; j++) w->x_prev[j] = w->q[j] + w->Px_[j] + w->Aty[j];
    w->info.dua_res = (w->s.scaling && !w->s.scaled_termination) ? w->cinv * scaled_norm_inf(w->Dinv, w->x_prev, n) : norm_inf(w->x_prev, n);
}

static double compute_rho_estimate(oracle_work *w) {
    int64_t n = w->n, m = w->m;
    double pri = norm_inf(w->z_prev, m), dua = norm_inf(w->x_prev, n);
    double pn = fmax(norm_inf(w->z, m), norm_inf(w->Ax_, m));
    pri /= (pn + 1e-10);
    double dn = fmax(fmax(norm_inf(w->q, n), norm_inf(w->Aty, n)), norm_inf(w->Px_, n));
    dua /= (dn + 1e-10);
    double r = w->s.rho * sqrt(pri / (dua + 1e-10));
    return fmin(fmax(r, RHO_MIN), RHO_MAX);
}

static int is_primal_infeasible(oracle_work *w, double eps) {
    int64_t n = w->n, m = w->m;
    int unscale = w->s.scaling && !w->s.scaled_termination;
    for (int64_t i = 0; i < m; i++) {
        if (w->u[i] > QP_INFTY * MIN_SCALING) {
            if (w->l[i] < -QP_INFTY * MIN_SCALING) w->delta_y[i] = 0.0;
            else w->delta_y[i] = fmin(w->delta_y[i], 0.0);
        } else if (w->l[i] < -QP_INFTY * MIN_SCALING) w->delta_y[i] = fmax(w->delta_y[i], 0.0);
    }
    double nd = unscale ? scaled_norm_inf(w->E, w->delta_y, m) : norm_inf(w->delta_y, m);
    if (nd > eps) {
        double lhs = 0;
        for (int64_t i = 0; i < m; i++) lhs += w->u[i] * fmax(w->delta_y[i], 0.0) + w->l[i] * fmin(w->delta_y[i], 0.0);
        if (lhs < -eps * nd) {
            mat_tpose_vec(n, w->Ap, w->Ai, w->Ax, w->delta_y, w->Atdy);
            if (unscale) for (int64_t j = 0; j < n; j++) w->Atdy[j] *= w->Dinv[j];
            return norm_inf(w->Atdy, n) < eps * nd;
        }
    }
    return 0;
}

static int is_dual_infeasible(oracle_work *w, double eps) {
    int64_t n = w->n, m = w->m;
    int unscale = w->s.scaling && !w->s.scaled_termination;
    double nd = unscale ? scaled_norm_inf(w->D, w->delta_x, n) : norm_inf(w->delta_x, n);
    double cs = unscale ? w->c : 1.0;
    if (nd > eps) {
        if (dot(w->q, w->delta_x, n) < -cs * eps * nd) {
            sym_mat_vec(n, w->Pp, w->Pi, w->Px, w->delta_x, w->Pdx);
            if (unscale) for (int64_t j = 0; j < n; j++) w->Pdx[j] *= w->Dinv[j];
            if (norm_inf(w->Pdx, n) < cs * eps * nd) {
                mat_vec(m, n, w->Ap, w->Ai, w->Ax, w->delta_x, w->Adx);
                if (unscale) for (int64_t i = 0; i < m; i++) w->Adx[i] *= w->Einv[i];
                for (int64_t i = 0; i < m; i++) {
                    if ((w->u[i] < QP_INFTY * MIN_SCALING && w->Adx[i] > eps * nd) ||
                        (w->l[i] > -QP_INFTY * MIN_SCALING && w->Adx[i] < -eps * nd)) return 0;
                }
                return 1;
            }
        }
    }
    return 0;
}

static int check_termination(oracle_work *w, int approximate) {
    int64_t n = w->n, m = w->m;
    double eps_abs = w->s.eps_abs, eps_rel = w->s.eps_rel, epi = w->s.eps_prim_inf, edi = w->s.eps_dual_inf;
    int unscale = w->s.scaling && !w->s.scaled_termination;
    if (w->info.pri_res > QP_INFTY || w->info.dua_res > QP_INFTY) { w->info.status = ST_NON_CVX; w->info.obj_val = NAN; return 1; }
    if (approximate) { eps_abs *= 10; eps_rel *= 10; epi *= 10; edi *= 10; }
    int pc = 0, dc = 0, pic = 0, dic = 0;
    if (m == 0) pc = 1;
    else {
        double mr = unscale ? fmax(scaled_norm_inf(w->Einv, w->z, m), scaled_norm_inf(w->Einv, w->Ax_, m))
                            : fmax(norm_inf(w->z, m), norm_inf(w->Ax_, m));
        double ep = eps_abs + eps_rel * mr;
        if (w->info.pri_res < ep) pc = 1; else pic = is_primal_infeasible(w, epi);
    }
    {
        double mr = unscale ? w->cinv * fmax(fmax(scaled_norm_inf(w->Dinv, w->q, n), scaled_norm_inf(w->Dinv, w->Aty, n)), scaled_norm_inf(w->Dinv, w->Px_, n))
                            : fmax(fmax(norm_inf(w->q, n), norm_inf(w->Aty, n)), norm_inf(w->Px_, n));
        double ed = eps_abs + eps_rel * mr;
        if (w->info.dua_res < ed) dc = 1; else dic = is_dual_infeasible(w, edi);
    }
    if (pc && dc) { w->info.status = approximate ? ST_SOLVED_INACCURATE : ST_SOLVED; return 1; }
    if (pic) { w->info.status = approximate ? ST_PRIMAL_INFEASIBLE_INACCURATE : ST_PRIMAL_INFEASIBLE; w->info.obj_val = QP_INFTY; return 1; }
    if (dic) { w->info.status = approximate ? ST_DUAL_INFEASIBLE_INACCURATE : ST_DUAL_INFEASIBLE; w->info.obj_val = -QP_INFTY; return 1; }
    return 0;
}

static void admm_iteration(oracle_work *w) {
    int64_t n = w->n, m = w->m;
    double alpha = w->s.alpha, sigma = w->s.sigma;
    /* swap: current iterate becomes "prev" */
    double *t = w->x; w->x = w->x_prev; w->x_prev = t;
    t = w->z; w->z = w->z_prev; w->z_prev = t;
    /* KKT right-hand side and solve */
    for (int64_t j = 0; j < n; j++) w->rhs[w->iperm[j]] = sigma * w->x_prev[j] - w->q[j];
    for (int64_t i = 0; i < m; i++) w->rhs[w->iperm[n + i]] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
    ldl_solve(w, w->rhs);
    for (int64_t j = 0; j < n; j++) w->xz_tilde[j] = w->rhs[w->iperm[j]];
    for (int64_t i = 0; i < m; i++) {
        double nu = w->rhs[w->iperm[n + i]];
        w->xz_tilde[n + i] = w->z_prev[i] + w->rho_inv_vec[i] * (nu - w->y[i]);
    }
    for (int64_t j = 0; j < n; j++) { w->x[j] = alpha * w->xz_tilde[j] + (1.0 - alpha) * w->x_prev[j]; w->delta_x[j] = w->x[j] - w->x_prev[j]; }
    for (int64_t i = 0; i < m; i++) {
        double v = alpha * w->xz_tilde[n + i] + (1.0 - alpha) * w->z_prev[i] + w->rho_inv_vec[i] * w->y[i];
        w->z[i] = fmin(fmax(v, w->l[i]), w->u[i]);
    }
    for (int64_t i = 0; i < m; i++) {
        w->delta_y[i] = w->rho_vec[i] * (alpha * w->xz_tilde[n + i] + (1.0 - alpha) * w->z_prev[i] - w->z[i]);
        w->y[i] += w->delta_y[i];
    }
}

/* Run at most `iters` plain ADMM iterations without any termination logic (test helper for
 * iterate-level comparisons against the GPU kernels). */
void oracle_iterate(oracle_work *w, int iters) { for (int k = 0; k < iters; k++) admm_iteration(w); }

int oracle_solve(oracle_work *w, double *x_out, double *y_out, oracle_info *info_out) {
    int64_t n = w->n, m = w->m;
    int iter, can_check = 0, done = 0;
    w->info.rho_updates = 0;    /* per-solve counter (OSQP accumulates; we report per call) */
    if (!w->s.warm_start) { memset(w->x, 0, sizeof(double) * (size_t)n); memset(w->z, 0, sizeof(double) * (size_t)m); memset(w->y, 0, sizeof(double) * (size_t)m); }
    w->info.status = ST_UNSOLVED;
    for (iter = 1; iter <= w->s.max_iter; iter++) {
        admm_iteration(w);
        can_check = w->s.check_termination && (iter % w->s.check_termination == 0);
        if (can_check) { update_info(w); if (check_termination(w, 0)) { done = 1; break; } }
        if (w->s.adaptive_rho && w->s.adaptive_rho_interval && (iter % w->s.adaptive_rho_interval == 0)) {
            if (!can_check) update_info(w);
            double rn = compute_rho_estimate(w);
            w->info.rho_estimate = rn;
            if (rn > w->s.rho * w->s.adaptive_rho_tolerance || rn < w->s.rho / w->s.adaptive_rho_tolerance) {
                w->s.rho = rn;
                set_rho_vec(w, 0);
                update_kkt_rho(w);
                ldl_numeric(w);
                w->info.rho_updates++;
            }
        }
    }
    if (!done) {
        iter = w->s.max_iter;
        if (!can_check) update_info(w);
        if (!check_termination(w, 1)) w->info.status = ST_MAX_ITER_REACHED;
    }
    w->info.iter = iter;
    w->info.rho_estimate = compute_rho_estimate(w);
    int st = w->info.status;
    int has_solution = !(st == ST_PRIMAL_INFEASIBLE || st == ST_PRIMAL_INFEASIBLE_INACCURATE ||
                         st == ST_DUAL_INFEASIBLE || st == ST_DUAL_INFEASIBLE_INACCURATE || st == ST_NON_CVX);
    if (has_solution) {
        if (x_out) for (int64_t j = 0; j < n; j++) x_out[j] = w->D[j] * w->x[j];
        if (y_out) for (int64_t i = 0; i < m; i++) y_out[i] = w->cinv * w->E[i] * w->y[i];
    } else {
        if (x_out) for (int64_t j = 0; j < n; j++) x_out[j] = NAN;
        if (y_out) for (int64_t i = 0; i < m; i++) y_out[i] = NAN;
        /* cold start the next solve */
        memset(w->x, 0, sizeof(double) * (size_t)n); memset(w->z, 0, sizeof(double) * (size_t)m); memset(w->y, 0, sizeof(double) * (size_t)m);
    }
    if (info_out) *info_out = w->info;
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Closed-loop driver for bench.py's cpu_baseline leg: the caller loop of the reference
 * (examples/example_point_mass.py:88-101; pyMPC/mpc.py:688-692) around this solver, in C, so that no
 * interpreter time sits between the solver calls.  Per step:
 *     u  = first input of the last solution, or uref unless 'solved'          (mpc.py:301-304)
 *     x+ = Ad x + Bd u + noise_k
 *     l[:nx] = u[:nx] = -x+ ; Delta-u_0 bounds = Dumin/Dumax + u ; q_U[0:nu] = q0_U[0:nu] - QDu u
 *                                                  (mpc.py:404-408,441-444; constant xref, uref)
 *     osqp.update(q, l, u); osqp.solve()                                       (mpc.py:454,369)
 * q0 is the linear cost for u_{-1} = 0; l, u, q are the caller's full vectors (updated in place); xsol [n] holds the
 * last solution on entry and exit, *status its status.  Returns the seconds spent in update + solve only.
 * --------------------------------------------------------------------------------------------- */
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

double oracle_mpc_closed_loop(oracle_work *w, int nsteps, int nx, int nu, int Np, int Nc,
                              const double *Ad, const double *Bd, const double *QDu, const double *uref,
                              const double *Dumin, const double *Dumax, const double *q0,
                              double *q, double *l, double *u, double *x, double *xsol, int *status,
                              const double *noise, long long *iters_out, long long *unsolved_out) {
    const int64_t ou = (int64_t)(Np + 1) * nx, rdu = 2 * ou + (int64_t)Nc * nu;
    double *xn = (double *)malloc(sizeof(double) * (size_t)nx), *uk = (double *)malloc(sizeof(double) * (size_t)nu);
    double spent = 0.0;
    long long iters = 0, unsolved = 0;
    for (int k = 0; k < nsteps; k++) {
        for (int j = 0; j < nu; j++) uk[j] = (*status == ST_SOLVED) ? xsol[ou + j] : uref[j];
        for (int i = 0; i < nx; i++) {
            double a = noise ? noise[(size_t)k * nx + i] : 0.0;
            for (int j = 0; j < nx; j++) a += Ad[i * nx + j] * x[j];
            for (int j = 0; j < nu; j++) a += Bd[i * nu + j] * uk[j];
            xn[i] = a;
        }
        memcpy(x, xn, sizeof(double) * (size_t)nx);
        for (int i = 0; i < nx; i++) { l[i] = -x[i]; u[i] = -x[i]; }
        for (int j = 0; j < nu; j++) {
            l[rdu + j] = Dumin[j] + uk[j]; u[rdu + j] = Dumax[j] + uk[j];
            double a = 0.0;
            for (int t = 0; t < nu; t++) a += QDu[j * nu + t] * uk[t];
            q[ou + j] = q0[ou + j] - a;
        }
        oracle_info info;
        double t0 = now_s();
        oracle_update(w, q, l, u);
        oracle_solve(w, xsol, NULL, &info);
        spent += now_s() - t0;
        *status = info.status;
        iters += info.iter;
        if (info.status != ST_SOLVED) unsolved++;
    }
    free(xn); free(uk);
    if (iters_out) *iters_out = iters;
    if (unsolved_out) *unsolved_out = unsolved;
    return spent;
}

/*
 * mpcqp_cpu.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * libmpcqp_cpu.so: the C ABI of include/mpcqp.h (what pympc_amd binds on the GPU) answered on the CPU by the oracle --
 * osqp_ref.c, the plain-C restatement of the solver pyMPC calls (pyMPC/mpc.py:241,266,369,454) -- behind a C restatement of the
 * reference's QP builder (pyMPC/mpc.py:456-608 setup, :386-454 update, bug for bug: the Delta-u stencil with its scalar offset,
 * the held last input of Nc < Np, SOFT_ON).  Same entry points, same argument meaning, same error codes, HOST pointers only.
 * Purpose: the Python host layer of the product -- MPCController, DeviceProblem, BatchMPCController, the CSC seam --
 * runs unchanged through this library in the CPU test-suite (tests/test_cpu_twin.py), and a maintainer can diff the two
 * libraries call by call.  Nothing under pympc_amd/ loads it; one instance at a time, one thread.
 * What has no meaning on a CPU (HIP streams, kernel names, stream-byte accounting, the KKT debug solve) returns
 * MPCQP_ERR_UNSUPPORTED or a neutral answer, as noted at each function.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mpcqp.h"

/* ---- the oracle (osqp_ref.c, compiled into the same library) ---------------------------------------------------------- */
typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, adaptive_rho_tolerance;
    int max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval, warm_start, scaled_termination;
} oracle_settings;
typedef struct { int status, iter, rho_updates; double obj_val, pri_res, dua_res, rho_estimate; } oracle_info;
typedef struct oracle_work oracle_work;
oracle_work *oracle_setup(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px, const int64_t *Ap, const int64_t *Ai,
                          const double *Ax, const double *q, const double *l, const double *u, const int64_t *perm, const oracle_settings *s);
void oracle_free(oracle_work *w);
int oracle_update(oracle_work *w, const double *q, const double *l, const double *u);
void oracle_warm_start(oracle_work *w, const double *x, const double *y);
int oracle_solve(oracle_work *w, double *x_out, double *y_out, oracle_info *info_out);
void oracle_iterate(oracle_work *w, int iters);
void oracle_get_scaling(const oracle_work *w, double *D, double *E, double *c);
void oracle_get_iterate(const oracle_work *w, double *x, double *z, double *y, double *rho);
int64_t oracle_nnzL(const oracle_work *w);
void oracle_set_tolerances(oracle_work *w, const oracle_settings *s);

#define QP_INFTY 1e30
static char g_err[512];
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }

typedef struct {
    int nx, nu, Np, Nc, N, n, m, n_x, n_u, ou, oe, rs, ri, rdu, soft;
} dims_t;

struct mpcqp_handle {
    int batch, is_setup, raw, xref_rows, generic;   /* generic: made by mpcqp_create_csc (the caller's matrices, any QP) */
    dims_t d;
    mpcqp_settings S;
    /* per instance */
    double *Ad, *Bd, *Qx, *QxN, *Qu, *QDu, *xmin, *xmax, *umin, *umax, *Dumin, *Dumax, *uref, *eps_feas;
    double *x0, *um1, *xref;                         /* [batch][nx], [nu], [N*nx] */
    double *q, *l, *u, *xs, *ys;                     /* current vectors and last solution */
    mpcqp_info *info;
    oracle_work **w;
    /* patterns (shared by the batch) */
    int64_t *Pp, *Pi, *Ap, *Ai, nnzP, nnzA;
    uint64_t stats[4];
};

static dims_t make_dims(int nx, int nu, int Np, int Nc, int soft) {
    dims_t d; d.nx = nx; d.nu = nu; d.Np = Np; d.Nc = Nc; d.N = Np + 1; d.soft = soft ? 1 : 0;
    d.n_x = d.N * nx; d.n_u = Nc * nu; d.n = (d.soft ? 2 : 1) * d.n_x + d.n_u; d.m = 2 * d.n_x + d.n_u + (Nc + 1) * nu;
    d.ou = d.n_x; d.oe = d.n_x + d.n_u; d.rs = d.n_x; d.ri = 2 * d.n_x; d.rdu = 2 * d.n_x + d.n_u;
    return d;
}

/* ---- the reference's matrices, column by column (CSC), pattern and values in one pass ------------------------------------
 * A (pyMPC/mpc.py:537-598): rows = dynamics | state box (x + eps if SOFT_ON) | input box | Delta-u (first nu rows: u_0, then
 * -I + super-diagonal at offset ONE SCALAR, mpc.py:570); the last input is held to the end of the horizon when Nc < Np (mpc.py:540-543).
 * P (mpc.py:482-531): blkdiag(I (x) Qx, QxN) | diag(iU) (x) Qu + iDu (x) QDu | I (x) Qeps -- the upper triangle, as osqp keeps it. */
typedef void (*emit_fn)(void *ctx, int64_t row, double val);
static void A_column(const dims_t *d, const double *Ad, const double *Bd, int j, emit_fn emit, void *ctx) {
    const int nx = d->nx, nu = d->nu;
    if (j < d->ou) {
        const int k = j / nx, i = j % nx;
        emit(ctx, j, -1.0);
        if (k < d->Np) for (int r = 0; r < nx; ++r) emit(ctx, (int64_t)(k + 1) * nx + r, Ad[r * nx + i]);
        emit(ctx, d->rs + j, 1.0);
    } else if (j < d->oe) {
        const int cc = j - d->ou, k = cc / nu, jj = cc % nu, s_end = (k == d->Nc - 1) ? d->Np : k + 1;
        for (int s = k + 1; s <= s_end; ++s) for (int r = 0; r < nx; ++r) emit(ctx, (int64_t)s * nx + r, Bd[r * nu + jj]);
        emit(ctx, d->ri + cc, 1.0);
        if (k == 0) emit(ctx, d->rdu + jj, 1.0);
        if (cc > 0) emit(ctx, d->rdu + nu + cc - 1, 1.0);
        emit(ctx, d->rdu + nu + cc, -1.0);
    } else emit(ctx, d->rs + (j - d->oe), 1.0);
}
static void P_column(const dims_t *d, const double *Qx, const double *QxN, const double *Qu, const double *QDu, double eps_feas, int c, emit_fn emit, void *ctx) {
    const int nx = d->nx, nu = d->nu;
    if (c < d->ou) {
        const int k = c / nx, l = c % nx; const double *Q = k < d->Np ? Qx : QxN;
        for (int i = 0; i <= l; ++i) emit(ctx, (int64_t)k * nx + i, Q[i * nx + l]);
    } else if (c < d->oe) {
        const int cc = c - d->ou, k = cc / nu, l = cc % nu;
        const double iu = (k == d->Nc - 1) ? (double)(d->Np - d->Nc + 1) : 1.0, dk = (k == d->Nc - 1) ? 1.0 : 2.0;
        if (k > 0) for (int jj = 0; jj < nu; ++jj) emit(ctx, d->ou + (int64_t)(k - 1) * nu + jj, -QDu[jj * nu + l]);
        for (int jj = 0; jj <= l; ++jj) emit(ctx, d->ou + (int64_t)k * nu + jj, iu * Qu[jj * nu + l] + dk * QDu[jj * nu + l]);
    } else emit(ctx, c, eps_feas);
}
typedef struct { int64_t *idx; double *val; int64_t cnt; } sink_t;
static void count_emit(void *c, int64_t r, double v) { (void)r; (void)v; ((sink_t *)c)->cnt++; }
static void store_emit(void *c, int64_t r, double v) { sink_t *s = (sink_t *)c; if (s->idx) s->idx[s->cnt] = r; if (s->val) s->val[s->cnt] = v; s->cnt++; }

/* q (mpc.py:411-452 / 489-526; the slack part is zero), l and u (mpc.py:551-580, 404-408) of one instance */
static void build_vectors(const mpcqp_handle *h, int b, double *q, double *l, double *u) {
    const dims_t *d = &h->d; const int nx = d->nx, nu = d->nu;
    const double *Qx = h->Qx + (size_t)b * nx * nx, *QxN = h->QxN + (size_t)b * nx * nx, *Qu = h->Qu + (size_t)b * nu * nu, *QDu = h->QDu + (size_t)b * nu * nu;
    const double *xref = h->xref + (size_t)b * d->N * nx, *uref = h->uref + (size_t)b * nu, *um1 = h->um1 + (size_t)b * nu, *x0 = h->x0 + (size_t)b * nx;
    for (int j = 0; j < d->n; ++j) q[j] = 0.0;
    for (int k = 0; k < d->N; ++k) {
        const double *Q = k < d->Np ? Qx : QxN, *xr = h->xref_rows == 1 ? xref : xref + (size_t)k * nx;
        for (int i = 0; i < nx; ++i) {
            double acc = 0.0;
            if (h->xref_rows == 1) for (int t = 0; t < nx; ++t) acc += Q[i * nx + t] * xr[t];
            else for (int t = 0; t < nx; ++t) acc += xr[t] * Q[t * nx + i];
            q[k * nx + i] = -acc;
        }
    }
    for (int k = 0; k < d->Nc; ++k) {
        const double iu = (k == d->Nc - 1) ? (double)(d->Np - d->Nc + 1) : 1.0;
        for (int jj = 0; jj < nu; ++jj) {
            double a = 0.0, dd = 0.0;
            for (int t = 0; t < nu; ++t) a += Qu[jj * nu + t] * uref[t];
            double acc = iu * (-a);
            if (k == 0) { for (int t = 0; t < nu; ++t) dd += QDu[jj * nu + t] * um1[t]; acc += -dd; }
            q[d->ou + k * nu + jj] = acc;
        }
    }
    const double *xmin = h->xmin + (size_t)b * nx, *xmax = h->xmax + (size_t)b * nx, *umin = h->umin + (size_t)b * nu, *umax = h->umax + (size_t)b * nu;
    const double *Dumin = h->Dumin + (size_t)b * nu, *Dumax = h->Dumax + (size_t)b * nu;
    for (int r = 0; r < d->m; ++r) {
        double lo, hi;
        if (r < d->rs) lo = hi = r < nx ? -x0[r] : 0.0;
        else if (r < d->ri) { lo = xmin[(r - d->rs) % nx]; hi = xmax[(r - d->rs) % nx]; }
        else if (r < d->rdu) { lo = umin[(r - d->ri) % nu]; hi = umax[(r - d->ri) % nu]; }
        else { const int rr = r - d->rdu, jj = rr % nu; lo = Dumin[jj]; hi = Dumax[jj]; if (rr < nu) { lo += um1[jj]; hi += um1[jj]; } }
        l[r] = lo < -QP_INFTY ? -QP_INFTY : lo; u[r] = hi > QP_INFTY ? QP_INFTY : hi;
    }
}

/* Ordering of the quasi-definite KKT matrix [P + sigma I, A'; A, -1/rho] for the LDL' without pivoting (perm[new] = old): the
 * constraint rows first -- pivots -1/rho, and what they leave on the variables is P + sigma I + A' rho A, the positive definite
 * block-tridiagonal matrix the GPU factors too -- then the variables stage by stage (banded fill).  (Variables first would take
 * pivots of size sigma = 1e-6 wherever P has a zero on its diagonal: six digits lost.) */
static int64_t *stage_ordering(const dims_t *d) {
    const int n = d->n, nk = d->n + d->m, nx = d->nx, nu = d->nu;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nk); int p = 0;
    for (int k = 0; k < d->N; ++k) {
        for (int i = 0; i < nx; ++i) perm[p++] = n + k * nx + i;
        for (int i = 0; i < nx; ++i) perm[p++] = n + d->rs + k * nx + i;
        if (k < d->Nc) { for (int j = 0; j < nu; ++j) perm[p++] = n + d->ri + k * nu + j; for (int j = 0; j < nu; ++j) perm[p++] = n + d->rdu + nu + k * nu + j; }
        if (k == 0) for (int j = 0; j < nu; ++j) perm[p++] = n + d->rdu + j;
    }
    for (int k = 0; k < d->N; ++k) {
        if (d->soft) for (int i = 0; i < nx; ++i) perm[p++] = d->oe + k * nx + i;
        for (int i = 0; i < nx; ++i) perm[p++] = k * nx + i;
        if (k < d->Nc) for (int j = 0; j < nu; ++j) perm[p++] = d->ou + k * nu + j;
    }
    return perm;
}

static oracle_settings to_oracle(const mpcqp_settings *s) {
    oracle_settings o; o.rho = s->rho; o.sigma = s->sigma; o.alpha = s->alpha; o.eps_abs = s->eps_abs; o.eps_rel = s->eps_rel;
    o.eps_prim_inf = s->eps_prim_inf; o.eps_dual_inf = s->eps_dual_inf; o.adaptive_rho_tolerance = s->adaptive_rho_tolerance;
    o.max_iter = s->max_iter; o.check_termination = s->check_termination; o.scaling = s->scaling; o.adaptive_rho = s->adaptive_rho;
    o.adaptive_rho_interval = s->adaptive_rho_interval; o.warm_start = s->warm_start; o.scaled_termination = 0;
    return o;
}

/* ---- ABI ------------------------------------------------------------------------------------------------------------- */
void mpcqp_default_settings(mpcqp_settings *s) {
    s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6; s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
    s->adaptive_rho_tolerance = 5.0; s->max_iter = 4000; s->check_termination = 25; s->scaling = 10;
    s->adaptive_rho = 1; s->adaptive_rho_interval = 0; s->warm_start = 1; s->soft_constraints = 1;
    s->backend = 0; s->tuning = 0;      /* (one backend here: the oracle; the fields are carried for layout compatibility) */
}
const char *mpcqp_status_string(int status) {
    switch (status) {
        case MPCQP_SOLVED: return "solved"; case MPCQP_SOLVED_INACCURATE: return "solved inaccurate";
        case MPCQP_MAX_ITER_REACHED: return "maximum iterations reached"; case MPCQP_PRIMAL_INFEASIBLE: return "primal infeasible";
        case MPCQP_PRIMAL_INFEASIBLE_INACCURATE: return "primal infeasible inaccurate"; case MPCQP_DUAL_INFEASIBLE: return "dual infeasible";
        case MPCQP_DUAL_INFEASIBLE_INACCURATE: return "dual infeasible inaccurate"; case MPCQP_NON_CVX: return "problem non convex";
        default: return "unsolved";
    }
}
const char *mpcqp_last_error(void) { return g_err; }
int mpcqp_device_count(void) { return 1; }          /* "the CPU" */

static double *dcalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static mpcqp_handle *new_handle(int batch, dims_t d, const mpcqp_settings *s) {
    mpcqp_handle *h = (mpcqp_handle *)calloc(1, sizeof(*h));
    h->batch = batch; h->d = d; h->xref_rows = 1;
    if (s) h->S = *s; else mpcqp_default_settings(&h->S);
    const size_t B = (size_t)batch, nx = d.nx, nu = d.nu;
    h->Ad = dcalloc(B * nx * nx); h->Bd = dcalloc(B * nx * nu); h->Qx = dcalloc(B * nx * nx); h->QxN = dcalloc(B * nx * nx);
    h->Qu = dcalloc(B * nu * nu); h->QDu = dcalloc(B * nu * nu); h->xmin = dcalloc(B * nx); h->xmax = dcalloc(B * nx);
    h->umin = dcalloc(B * nu); h->umax = dcalloc(B * nu); h->Dumin = dcalloc(B * nu); h->Dumax = dcalloc(B * nu);
    h->uref = dcalloc(B * nu); h->eps_feas = dcalloc(B); h->x0 = dcalloc(B * nx); h->um1 = dcalloc(B * nu); h->xref = dcalloc(B * d.N * nx);
    h->q = dcalloc(B * d.n); h->l = dcalloc(B * d.m); h->u = dcalloc(B * d.m); h->xs = dcalloc(B * d.n); h->ys = dcalloc(B * d.m);
    h->info = (mpcqp_info *)calloc(B, sizeof(mpcqp_info)); h->w = (oracle_work **)calloc(B, sizeof(oracle_work *));
    for (size_t b = 0; b < B; ++b) h->info[b].status = MPCQP_UNSOLVED;
    return h;
}
int mpcqp_create(mpcqp_handle **out, int device, int batch, int nx, int nu, int Np, int Nc, const mpcqp_settings *s) {
    (void)device;
    if (!out || batch < 1 || nx < 1 || nu < 1 || Np < 2 || Nc < 1 || Nc > Np) return fail(MPCQP_ERR_ARG, "mpcqp_create: bad dimensions");
    mpcqp_settings st; if (s) st = *s; else mpcqp_default_settings(&st);
    *out = new_handle(batch, make_dims(nx, nu, Np, Nc, st.soft_constraints), &st);
    return MPCQP_OK;
}
void mpcqp_destroy(mpcqp_handle *h) {
    if (!h) return;
    for (int b = 0; b < h->batch; ++b) if (h->w[b]) oracle_free(h->w[b]);
    void *p[] = {h->Ad, h->Bd, h->Qx, h->QxN, h->Qu, h->QDu, h->xmin, h->xmax, h->umin, h->umax, h->Dumin, h->Dumax, h->uref, h->eps_feas, h->x0, h->um1,
                 h->xref, h->q, h->l, h->u, h->xs, h->ys, h->info, h->w, h->Pp, h->Pi, h->Ap, h->Ai};
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i) free(p[i]);
    free(h);
}
int mpcqp_set_stream(mpcqp_handle *h, void *s) { (void)s; return h ? MPCQP_OK : fail(MPCQP_ERR_ARG, "null handle"); }      /* no streams on a CPU */
int mpcqp_synchronize(mpcqp_handle *h) { return h ? MPCQP_OK : fail(MPCQP_ERR_ARG, "null handle"); }

static void copy_rows(double *dst, const double *src, size_t batch, size_t w) { if (src) memcpy(dst, src, sizeof(double) * batch * w); }
static int step_upload(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows) {
    const dims_t *d = &h->d; const size_t B = (size_t)h->batch;
    copy_rows(h->x0, x0, B, d->nx); copy_rows(h->um1, um1, B, d->nu);
    if (xref) {
        if (xref_rows != 1 && xref_rows != d->N) return fail(MPCQP_ERR_ARG, "xref_rows must be 1 or Np+1");
        h->xref_rows = xref_rows;
        for (size_t b = 0; b < B; ++b) memcpy(h->xref + b * d->N * d->nx, xref + b * (size_t)xref_rows * d->nx, sizeof(double) * (size_t)xref_rows * d->nx);
    }
    return 0;
}

/* patterns of P (upper triangle) and A from the dimensions, values of instance b into Pv / Av */
static void build_patterns(mpcqp_handle *h) {
    const dims_t *d = &h->d; sink_t s;
    h->Pp = (int64_t *)calloc((size_t)d->n + 1, sizeof(int64_t)); h->Ap = (int64_t *)calloc((size_t)d->n + 1, sizeof(int64_t));
    for (int c = 0; c < d->n; ++c) {
        s.cnt = 0; P_column(d, h->Qx, h->QxN, h->Qu, h->QDu, h->eps_feas[0], c, count_emit, &s); h->Pp[c + 1] = h->Pp[c] + s.cnt;
        s.cnt = 0; A_column(d, h->Ad, h->Bd, c, count_emit, &s); h->Ap[c + 1] = h->Ap[c] + s.cnt;
    }
    h->nnzP = h->Pp[d->n]; h->nnzA = h->Ap[d->n];
    h->Pi = (int64_t *)malloc(sizeof(int64_t) * (size_t)h->nnzP); h->Ai = (int64_t *)malloc(sizeof(int64_t) * (size_t)h->nnzA);
    for (int c = 0; c < d->n; ++c) {
        s.idx = h->Pi + h->Pp[c]; s.val = NULL; s.cnt = 0; P_column(d, h->Qx, h->QxN, h->Qu, h->QDu, h->eps_feas[0], c, store_emit, &s);
        s.idx = h->Ai + h->Ap[c]; s.val = NULL; s.cnt = 0; A_column(d, h->Ad, h->Bd, c, store_emit, &s);
    }
}
static void build_values(const mpcqp_handle *h, int b, double *Pv, double *Av) {
    const dims_t *d = &h->d; const size_t nx = d->nx, nu = d->nu; sink_t s;
    for (int c = 0; c < d->n; ++c) {
        s.idx = NULL; s.val = Pv + h->Pp[c]; s.cnt = 0;
        P_column(d, h->Qx + b * nx * nx, h->QxN + b * nx * nx, h->Qu + b * nu * nu, h->QDu + b * nu * nu, h->eps_feas[b], c, store_emit, &s);
        s.idx = NULL; s.val = Av + h->Ap[c]; s.cnt = 0;
        A_column(d, h->Ad + b * nx * nx, h->Bd + b * nx * nu, c, store_emit, &s);
    }
}
static int factor_all_instances(mpcqp_handle *h, const double *Pval, const double *Aval) {
    /* Pval / Aval: [batch][nnz] values in the order of the handle's patterns, or NULL = build them from the model */
    const dims_t *d = &h->d; const oracle_settings os = to_oracle(&h->S);
    int64_t *perm;
    if (h->generic) {          /* the caller's matrices: constraint rows first, then the variables, each in their own order (see stage_ordering) */
        perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(d->n + d->m));
        for (int i = 0; i < d->m; ++i) perm[i] = d->n + i;
        for (int j = 0; j < d->n; ++j) perm[d->m + j] = j;
    } else perm = stage_ordering(d);
    double *Pv = (double *)malloc(sizeof(double) * (size_t)(h->nnzP ? h->nnzP : 1)), *Av = (double *)malloc(sizeof(double) * (size_t)(h->nnzA ? h->nnzA : 1));
    int rc = MPCQP_OK;
    for (int b = 0; b < h->batch && rc == MPCQP_OK; ++b) {
        if (Pval) { memcpy(Pv, Pval + (size_t)b * h->nnzP, sizeof(double) * (size_t)h->nnzP); memcpy(Av, Aval + (size_t)b * h->nnzA, sizeof(double) * (size_t)h->nnzA); }
        else build_values(h, b, Pv, Av);
        if (h->w[b]) oracle_free(h->w[b]);
        h->w[b] = oracle_setup(d->n, d->m, h->Pp, h->Pi, Pv, h->Ap, h->Ai, Av, h->q + (size_t)b * d->n, h->l + (size_t)b * d->m, h->u + (size_t)b * d->m, perm, &os);
        if (!h->w[b]) rc = fail(MPCQP_ERR_ARG, "oracle_setup failed (singular KKT matrix?)");
        h->info[b].status = MPCQP_UNSOLVED; h->info[b].iter = 0; h->info[b].rho_updates = 0; h->info[b].rho = h->S.rho;
    }
    free(perm); free(Pv); free(Av);
    if (rc == MPCQP_OK) h->is_setup = 1;
    return rc;
}

static void upload_model(mpcqp_handle *h, const mpcqp_model *M, int with_bounds) {
    const dims_t *d = &h->d; const size_t B = (size_t)h->batch, nx = d->nx, nu = d->nu;
    copy_rows(h->Ad, M->Ad, B, nx * nx); copy_rows(h->Bd, M->Bd, B, nx * nu); copy_rows(h->Qx, M->Qx, B, nx * nx); copy_rows(h->QxN, M->QxN, B, nx * nx);
    copy_rows(h->Qu, M->Qu, B, nu * nu); copy_rows(h->QDu, M->QDu, B, nu * nu); copy_rows(h->eps_feas, M->eps_feas, B, 1); copy_rows(h->uref, M->uref, B, nu);
    if (with_bounds) { copy_rows(h->xmin, M->xmin, B, nx); copy_rows(h->xmax, M->xmax, B, nx); copy_rows(h->umin, M->umin, B, nu); copy_rows(h->umax, M->umax, B, nu);
                       copy_rows(h->Dumin, M->Dumin, B, nu); copy_rows(h->Dumax, M->Dumax, B, nu); }
}

int mpcqp_setup(mpcqp_handle *h, const mpcqp_model *M, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h || !M || !x0 || !um1 || !xref) return fail(MPCQP_ERR_ARG, "mpcqp_setup: null argument");
    if (!M->Ad || !M->Bd || !M->Qx || !M->QxN || !M->Qu || !M->QDu || !M->xmin || !M->xmax || !M->umin || !M->umax || !M->Dumin || !M->Dumax || !M->uref || !M->eps_feas)
        return fail(MPCQP_ERR_ARG, "mpcqp_setup: null model field");
    if (h->generic) return fail(MPCQP_ERR_STATE, "mpcqp_setup on a handle made by mpcqp_create_csc");
    upload_model(h, M, 1);
    int rc = step_upload(h, x0, um1, xref, xref_rows); if (rc) return rc;
    if (!h->Pp) build_patterns(h);
    h->raw = 0;
    for (int b = 0; b < h->batch; ++b) build_vectors(h, b, h->q + (size_t)b * h->d.n, h->l + (size_t)b * h->d.m, h->u + (size_t)b * h->d.m);
    return factor_all_instances(h, NULL, NULL);
}
int mpcqp_setup_qp(mpcqp_handle *h, const mpcqp_model *M, const double *q, const double *l, const double *u) {
    if (!h || !M || !q || !l || !u) return fail(MPCQP_ERR_ARG, "mpcqp_setup_qp: null argument");
    if (!M->Ad || !M->Bd || !M->Qx || !M->QxN || !M->Qu || !M->QDu || !M->eps_feas) return fail(MPCQP_ERR_ARG, "mpcqp_setup_qp: null model field");
    if (h->generic) return fail(MPCQP_ERR_STATE, "mpcqp_setup_qp on a handle made by mpcqp_create_csc");
    upload_model(h, M, 0);
    if (!h->Pp) build_patterns(h);
    h->raw = 1;
    memcpy(h->q, q, sizeof(double) * (size_t)h->batch * h->d.n); memcpy(h->l, l, sizeof(double) * (size_t)h->batch * h->d.m); memcpy(h->u, u, sizeof(double) * (size_t)h->batch * h->d.m);
    return factor_all_instances(h, NULL, NULL);
}
static int push_vectors(mpcqp_handle *h, int with_q, int with_lu) {
    for (int b = 0; b < h->batch; ++b)
        if (oracle_update(h->w[b], with_q ? h->q + (size_t)b * h->d.n : NULL, with_lu ? h->l + (size_t)b * h->d.m : NULL, with_lu ? h->u + (size_t)b * h->d.m : NULL))
            return fail(MPCQP_ERR_ARG, "lower bound must be lower than or equal to upper bound");
    return MPCQP_OK;
}
int mpcqp_update(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_update before mpcqp_setup");
    if (h->generic) return fail(MPCQP_ERR_STATE, "mpcqp_update on a handle made by mpcqp_create_csc (use mpcqp_update_vectors)");
    int rc = step_upload(h, x0, um1, xref, xref_rows); if (rc) return rc;
    h->raw = 0;
    for (int b = 0; b < h->batch; ++b) build_vectors(h, b, h->q + (size_t)b * h->d.n, h->l + (size_t)b * h->d.m, h->u + (size_t)b * h->d.m);
    return push_vectors(h, 1, 1);
}
int mpcqp_update_vectors(mpcqp_handle *h, const double *q, const double *l, const double *u) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_update_vectors before setup");
    if ((l == NULL) != (u == NULL)) return fail(MPCQP_ERR_ARG, "mpcqp_update_vectors: give l and u together");
    h->raw = 1;
    if (q) memcpy(h->q, q, sizeof(double) * (size_t)h->batch * h->d.n);
    if (l) { memcpy(h->l, l, sizeof(double) * (size_t)h->batch * h->d.m); memcpy(h->u, u, sizeof(double) * (size_t)h->batch * h->d.m); }
    return push_vectors(h, q != NULL, l != NULL);
}

int mpcqp_create_csc(mpcqp_handle **out, int device, int batch, int n, int m, const int64_t *P_colptr, const int32_t *P_rowidx,
                     const int64_t *A_colptr, const int32_t *A_rowidx, int nx_hint, int nu_hint, const mpcqp_settings *s) {
    (void)device; (void)nx_hint; (void)nu_hint;
    if (!out || !P_colptr || !P_rowidx || !A_colptr || !A_rowidx || n < 1 || m < 1 || batch < 1) return fail(MPCQP_ERR_ARG, "mpcqp_create_csc: null argument");
    /* the CPU solver works on the matrices as they are: ANY convex QP is accepted (the GPU library accepts pyMPC's only) */
    dims_t d; memset(&d, 0, sizeof(d)); d.n = n; d.m = m; d.nx = d.nu = 1; d.N = 1; d.soft = 1;
    mpcqp_handle *h = new_handle(batch, d, s);
    h->generic = 1;
    h->Pp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1)); h->Ap = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    memcpy(h->Pp, P_colptr, sizeof(int64_t) * (size_t)(n + 1)); memcpy(h->Ap, A_colptr, sizeof(int64_t) * (size_t)(n + 1));
    /* keep the upper triangle of P only (osqp does): entries below the diagonal are dropped when the values arrive; here all are kept and masked there */
    h->nnzP = h->Pp[n]; h->nnzA = h->Ap[n];
    h->Pi = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h->nnzP ? h->nnzP : 1)); h->Ai = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h->nnzA ? h->nnzA : 1));
    for (int64_t p = 0; p < h->nnzP; ++p) h->Pi[p] = P_rowidx[p];
    for (int64_t p = 0; p < h->nnzA; ++p) h->Ai[p] = A_rowidx[p];
    *out = h;
    return MPCQP_OK;
}
int mpcqp_setup_csc(mpcqp_handle *h, const double *P_val, const double *A_val, const double *q, const double *l, const double *u) {
    if (!h || !P_val || !A_val || !q || !l || !u) return fail(MPCQP_ERR_ARG, "mpcqp_setup_csc: null argument");
    if (!h->generic) return fail(MPCQP_ERR_STATE, "mpcqp_setup_csc: the handle was not made by mpcqp_create_csc");
    const size_t B = (size_t)h->batch;
    /* triu(P): zero the strictly lower entries of a full symmetric input (a structural zero is harmless to the solver) */
    double *Pv = (double *)malloc(sizeof(double) * B * (size_t)(h->nnzP ? h->nnzP : 1));
    memcpy(Pv, P_val, sizeof(double) * B * (size_t)h->nnzP);
    for (int c = 0; c < h->d.n; ++c) for (int64_t p = h->Pp[c]; p < h->Pp[c + 1]; ++p) if (h->Pi[p] > c) for (size_t b = 0; b < B; ++b) Pv[b * h->nnzP + p] = 0.0;
    memcpy(h->q, q, sizeof(double) * B * h->d.n); memcpy(h->l, l, sizeof(double) * B * h->d.m); memcpy(h->u, u, sizeof(double) * B * h->d.m);
    h->raw = 1;
    const int rc = factor_all_instances(h, Pv, A_val);
    free(Pv);
    return rc;
}

int mpcqp_warm_start(mpcqp_handle *h, const double *x, const double *y) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "warm_start before setup");
    for (int b = 0; b < h->batch; ++b) oracle_warm_start(h->w[b], x ? x + (size_t)b * h->d.n : NULL, y ? y + (size_t)b * h->d.m : NULL);
    return MPCQP_OK;
}
int mpcqp_update_settings(mpcqp_handle *h, const mpcqp_settings *s) {
    if (!h || !s) return fail(MPCQP_ERR_ARG, "null argument");
    const double rho = h->S.rho, sigma = h->S.sigma; const int scaling = h->S.scaling, soft = h->S.soft_constraints;
    h->S = *s; h->S.rho = rho; h->S.sigma = sigma; h->S.scaling = scaling; h->S.soft_constraints = soft;
    const oracle_settings os = to_oracle(&h->S);
    for (int b = 0; b < h->batch; ++b) if (h->w[b]) oracle_set_tolerances(h->w[b], &os);
    return MPCQP_OK;
}
int mpcqp_solve(mpcqp_handle *h) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "solve before mpcqp_setup");
    for (int b = 0; b < h->batch; ++b) {
        oracle_info oi;
        oracle_solve(h->w[b], h->xs + (size_t)b * h->d.n, h->ys + (size_t)b * h->d.m, &oi);
        mpcqp_info *inf = &h->info[b];
        inf->status = oi.status; inf->iter = oi.iter; inf->rho_updates = oi.rho_updates; inf->reserved = 0;
        inf->obj_val = oi.obj_val; inf->pri_res = oi.pri_res; inf->dua_res = oi.dua_res; inf->rho = oi.rho_estimate;
        h->stats[0] += (uint64_t)oi.iter; h->stats[1] += (uint64_t)((oi.iter + h->S.check_termination - 1) / (h->S.check_termination ? h->S.check_termination : 1));
        h->stats[2] += (uint64_t)oi.rho_updates; h->stats[3] += 1;
    }
    return MPCQP_OK;
}
int mpcqp_get_solution(mpcqp_handle *h, double *x, double *y, mpcqp_info *info) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (x) memcpy(x, h->xs, sizeof(double) * (size_t)h->batch * h->d.n);
    if (y) memcpy(y, h->ys, sizeof(double) * (size_t)h->batch * h->d.m);
    if (info) memcpy(info, h->info, sizeof(mpcqp_info) * (size_t)h->batch);
    return MPCQP_OK;
}
int mpcqp_get_u0(mpcqp_handle *h, double *u0) {
    if (!h || !u0) return fail(MPCQP_ERR_ARG, "null argument");
    for (int b = 0; b < h->batch; ++b) memcpy(u0 + (size_t)b * h->d.nu, h->xs + (size_t)b * h->d.n + h->d.ou, sizeof(double) * (size_t)h->d.nu);
    return MPCQP_OK;
}
/* output() of mpc.py:271-336: the first input of the solution if 'solved', else u_failure (= uref) */
static void output_u(const mpcqp_handle *h, int b, double *u) {
    for (int j = 0; j < h->d.nu; ++j) u[j] = h->info[b].status == MPCQP_SOLVED ? h->xs[(size_t)b * h->d.n + h->d.ou + j] : h->uref[(size_t)b * h->d.nu + j];
}
int mpcqp_mpc_step(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows, double *u_out) {
    if (!h || !x0 || !u_out) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_step: null argument");
    int rc = mpcqp_update(h, x0, um1, xref, xref_rows); if (rc) return rc;
    if ((rc = mpcqp_solve(h))) return rc;
    for (int b = 0; b < h->batch; ++b) { output_u(h, b, u_out + (size_t)b * h->d.nu); memcpy(h->um1 + (size_t)b * h->d.nu, u_out + (size_t)b * h->d.nu, sizeof(double) * (size_t)h->d.nu); }
    return MPCQP_OK;
}
int mpcqp_step_host(mpcqp_handle *h, const double *x0, const double *um1, const double *xref, int xref_rows, double *x, double *y, mpcqp_info *info) {
    int rc = mpcqp_update(h, x0, um1, xref, xref_rows); if (rc) return rc;
    if ((rc = mpcqp_solve(h))) return rc;
    return mpcqp_get_solution(h, x, y, info);
}
int mpcqp_mpc_loop(mpcqp_handle *h, int nsteps, const mpcqp_loop *io) {
    if (!h || !io || nsteps < 1) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: bad argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_loop before mpcqp_setup");
    if (h->raw) return fail(MPCQP_ERR_STATE, "mpcqp_mpc_loop: the handle holds raw q, l, u (mpcqp_update_vectors); call mpcqp_update first");
    if (io->ny) return fail(MPCQP_ERR_UNSUPPORTED, "mpcqp_mpc_loop: output feedback is not implemented in the CPU twin");
    if ((io->Ap == NULL) != (io->Bp == NULL)) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: give both Ap and Bp or neither");
    const dims_t *d = &h->d; const size_t B = (size_t)h->batch, nx = d->nx, nu = d->nu;
    const int rows = io->xref_traj ? (io->xref_rows ? io->xref_rows : h->xref_rows) : h->xref_rows;
    if (io->xref_traj && rows != 1 && rows != d->N) return fail(MPCQP_ERR_ARG, "mpcqp_mpc_loop: xref_rows must be 0 (as last uploaded), 1 or Np+1");
    double *xn = dcalloc(B * nx), *un = dcalloc(B * nu);
    for (int k = 0; k < nsteps; ++k) {
        for (size_t b = 0; b < B; ++b) {
            double *u = un + b * nu; const double *x = h->x0 + b * nx;
            output_u(h, (int)b, u);
            const double *Ap = io->Ap ? io->Ap + b * nx * nx : h->Ad + b * nx * nx, *Bp = io->Bp ? io->Bp + b * nx * nu : h->Bd + b * nx * nu;
            for (size_t i = 0; i < nx; ++i) {
                double acc = 0.0;
                for (size_t j = 0; j < nx; ++j) acc += Ap[i * nx + j] * x[j];
                for (size_t j = 0; j < nu; ++j) acc += Bp[i * nu + j] * u[j];
                xn[b * nx + i] = acc + (io->w ? io->w[((size_t)k * B + b) * nx + i] : 0.0);
            }
            if (io->x_traj) memcpy(io->x_traj + ((size_t)k * B + b) * nx, x, sizeof(double) * nx);
            if (io->u_traj) memcpy(io->u_traj + ((size_t)k * B + b) * nu, u, sizeof(double) * nu);
        }
        int rc = mpcqp_update(h, xn, un, io->xref_traj ? io->xref_traj + (size_t)k * B * rows * nx : NULL, rows);
        if (rc == MPCQP_OK) rc = mpcqp_solve(h);
        if (rc) { free(xn); free(un); return rc; }
        for (size_t b = 0; b < B; ++b) {
            if (io->status_traj) io->status_traj[(size_t)k * B + b] = h->info[b].status;
            if (io->iter_traj) io->iter_traj[(size_t)k * B + b] = h->info[b].iter;
        }
    }
    if (io->x_traj) memcpy(io->x_traj + (size_t)nsteps * B * nx, h->x0, sizeof(double) * B * nx);
    free(xn); free(un);
    return MPCQP_OK;
}
int mpcqp_mpc_run(mpcqp_handle *h, int nsteps, const double *w, const double *Ap, const double *Bp, double *x_traj, double *u_traj, int32_t *status_traj, int32_t *iter_traj) {
    mpcqp_loop io; memset(&io, 0, sizeof(io));
    io.w = w; io.Ap = Ap; io.Bp = Bp; io.x_traj = x_traj; io.u_traj = u_traj; io.status_traj = status_traj; io.iter_traj = iter_traj;
    return mpcqp_mpc_loop(h, nsteps, &io);
}
int mpcqp_get_stats(mpcqp_handle *h, uint64_t *out4, int reset) {
    if (!h || !out4) return fail(MPCQP_ERR_ARG, "null argument");
    memcpy(out4, h->stats, sizeof(h->stats)); if (reset) memset(h->stats, 0, sizeof(h->stats));
    return MPCQP_OK;
}
int mpcqp_get_launch_times(mpcqp_handle *h, uint64_t *out, int nsteps) {      /* no launches here: zeros */
    if (!h || !out || nsteps < 0 || nsteps > 64) return fail(MPCQP_ERR_ARG, "mpcqp_get_launch_times: bad argument"); memset(out, 0, (size_t)(2 + nsteps) * sizeof(uint64_t) * (size_t)h->batch); return MPCQP_OK;
}
int mpcqp_profile(mpcqp_handle *h, int enable, double *run_ms, int64_t *run_launches, int reset) {      /* no kernels to time */
    (void)enable; (void)reset; if (!h) return fail(MPCQP_ERR_ARG, "null handle"); if (run_ms) *run_ms = 0.0; if (run_launches) *run_launches = 0; return MPCQP_OK;
}
int mpcqp_get_shape(mpcqp_handle *h, int *nx, int *nu, int *Np, int *Nc) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (h->generic) return fail(MPCQP_ERR_UNSUPPORTED, "the CPU twin does not read controller dimensions out of matrices (it solves them as they are)");
    if (nx) *nx = h->d.nx; if (nu) *nu = h->d.nu; if (Np) *Np = h->d.Np; if (Nc) *Nc = h->d.Nc;
    return MPCQP_OK;
}
int mpcqp_get_dims(mpcqp_handle *h, int *n, int *m, int64_t *factor_doubles, int64_t *nnzL) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (n) *n = h->d.n; if (m) *m = h->d.m;
    const int64_t nz = h->w && h->w[0] ? oracle_nnzL(h->w[0]) : 0;
    if (factor_doubles) *factor_doubles = nz; if (nnzL) *nnzL = nz + h->d.n + h->d.m;
    return MPCQP_OK;
}
int mpcqp_get_stream_bytes(mpcqp_handle *h, int64_t *a, int64_t *b, int64_t *c) { (void)a; (void)b; (void)c; (void)h; return fail(MPCQP_ERR_UNSUPPORTED, "no kernels on a CPU"); }
int mpcqp_get_work(mpcqp_handle *h, int64_t *v) { (void)h; (void)v; return fail(MPCQP_ERR_UNSUPPORTED, "no matrix cores on a CPU"); }
int mpcqp_get_occupancy(mpcqp_handle *h, int *w, int *c, int *t) { (void)h; if (w) *w = 1; if (c) *c = 1; if (t) *t = 1; return MPCQP_OK; }
int mpcqp_kernel_name(mpcqp_handle *h, int loop, char *buf, int buflen) { (void)h; (void)loop; if (buf && buflen > 0) snprintf(buf, (size_t)buflen, "cpu:osqp_ref"); return MPCQP_OK; }
int mpcqp_export_qp(mpcqp_handle *h, double *Pm, double *Am, double *q, double *l, double *u) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "export before setup");
    if (h->generic) return fail(MPCQP_ERR_UNSUPPORTED, "export of caller-supplied matrices");
    const dims_t *d = &h->d; const size_t n = d->n, m = d->m;
    double *Pv = (double *)malloc(sizeof(double) * (size_t)h->nnzP), *Av = (double *)malloc(sizeof(double) * (size_t)h->nnzA);
    for (int b = 0; b < h->batch; ++b) {
        build_values(h, b, Pv, Av);
        if (Pm) { double *o = Pm + (size_t)b * n * n; memset(o, 0, sizeof(double) * n * n);
                  for (size_t c = 0; c < n; ++c) for (int64_t p = h->Pp[c]; p < h->Pp[c + 1]; ++p) { o[(size_t)h->Pi[p] * n + c] = Pv[p]; o[c * n + (size_t)h->Pi[p]] = Pv[p]; } }
        if (Am) { double *o = Am + (size_t)b * m * n; memset(o, 0, sizeof(double) * m * n);
                  for (size_t c = 0; c < n; ++c) for (int64_t p = h->Ap[c]; p < h->Ap[c + 1]; ++p) o[(size_t)h->Ai[p] * n + c] = Av[p]; }
    }
    free(Pv); free(Av);
    if (q) memcpy(q, h->q, sizeof(double) * (size_t)h->batch * n);
    if (l && u) { memcpy(l, h->l, sizeof(double) * (size_t)h->batch * m); memcpy(u, h->u, sizeof(double) * (size_t)h->batch * m); }
    return MPCQP_OK;
}
int mpcqp_get_scaling(mpcqp_handle *h, double *D, double *E, double *c, double *rho) {
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    for (int b = 0; b < h->batch; ++b) {
        double cc; double *Dt = dcalloc((size_t)h->d.n), *Et = dcalloc((size_t)h->d.m), *xt = dcalloc((size_t)h->d.n), *zt = dcalloc((size_t)h->d.m), *yt = dcalloc((size_t)h->d.m), rh;
        oracle_get_scaling(h->w[b], Dt, Et, &cc); oracle_get_iterate(h->w[b], xt, zt, yt, &rh);
        if (D) memcpy(D + (size_t)b * h->d.n, Dt, sizeof(double) * (size_t)h->d.n);
        if (E) memcpy(E + (size_t)b * h->d.m, Et, sizeof(double) * (size_t)h->d.m);
        if (c) c[b] = cc; if (rho) rho[b] = rh;
        free(Dt); free(Et); free(xt); free(zt); free(yt);
    }
    return MPCQP_OK;
}
int mpcqp_debug_kkt_solve(mpcqp_handle *h, const double *rhs, double *sol) { (void)h; (void)rhs; (void)sol; return fail(MPCQP_ERR_UNSUPPORTED, "the CPU twin solves the full KKT system, not the reduced one"); }
int mpcqp_get_iterate(mpcqp_handle *h, double *x, double *z, double *y) {      /* unscaled units, like the GPU library */
    if (!h) return fail(MPCQP_ERR_ARG, "null handle");
    const size_t n = h->d.n, m = h->d.m;
    double *D = dcalloc(n), *E = dcalloc(m), *xs = dcalloc(n), *zs = dcalloc(m), *ys = dcalloc(m), cc, rho;
    for (int b = 0; b < h->batch; ++b) {
        oracle_get_scaling(h->w[b], D, E, &cc); oracle_get_iterate(h->w[b], xs, zs, ys, &rho);
        for (size_t j = 0; j < n; ++j) if (x) x[(size_t)b * n + j] = D[j] * xs[j];
        for (size_t i = 0; i < m; ++i) { if (z) z[(size_t)b * m + i] = zs[i] / E[i]; if (y) y[(size_t)b * m + i] = E[i] * ys[i] / cc; }
    }
    free(D); free(E); free(xs); free(zs); free(ys);
    return MPCQP_OK;
}
int mpcqp_iterate(mpcqp_handle *h, int iters) {
    if (!h || iters < 1) return fail(MPCQP_ERR_ARG, "mpcqp_iterate: bad argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "solve before mpcqp_setup");
    for (int b = 0; b < h->batch; ++b) oracle_iterate(h->w[b], iters);
    return MPCQP_OK;
}
/* the equality-constrained part by multiplier sweeps: with the settings the caller gives such a handle (alpha = 1, fixed rho, all
 * inequality bounds infinite) one ADMM iteration of the oracle IS one sweep; an instance stops once a sweep moves x by less than
 * tol * max(1, |x|); the four residual norms are evaluated on the instance's own P and A as the device does (k_eq_solve) */
int mpcqp_eq_solve(mpcqp_handle *h, int sweeps, int cold, double tol, double *res) {
    if (!h || sweeps < 0) return fail(MPCQP_ERR_ARG, "mpcqp_eq_solve: bad argument");
    if (!h->is_setup) return fail(MPCQP_ERR_STATE, "mpcqp_eq_solve before mpcqp_setup");
    const size_t n = h->d.n, m = h->d.m;
    double *D = dcalloc(n), *E = dcalloc(m), *xs = dcalloc(n), *zs = dcalloc(m), *ys = dcalloc(m), *xp = dcalloc(n), cc, rho;
    for (int b = 0; b < h->batch; ++b) {
        if (cold) { memset(xs, 0, sizeof(double) * n); memset(ys, 0, sizeof(double) * m); oracle_warm_start(h->w[b], xs, ys); }
        oracle_get_scaling(h->w[b], D, E, &cc);
        int done = 0, settled = 0;
        for (int k = 0; k < sweeps; ++k) {
            oracle_get_iterate(h->w[b], xp, zs, ys, &rho);
            oracle_iterate(h->w[b], 1);
            oracle_get_iterate(h->w[b], xs, zs, ys, &rho);
            done = k + 1;
            double dm = 0.0, xm = 0.0;
            for (size_t j = 0; j < n; ++j) { dm = fmax(dm, fabs(D[j] * (xs[j] - xp[j]))); xm = fmax(xm, fabs(D[j] * xs[j])); }
            if (k >= 2 && dm <= tol * fmax(1.0, xm)) { settled = 1; break; }      /* (an ADMM iteration sees the right-hand side b only through z: from a cold start the first one moves nothing) */
        }
        oracle_get_iterate(h->w[b], xs, zs, ys, &rho);
        for (size_t j = 0; j < n; ++j) h->xs[(size_t)b * n + j] = D[j] * xs[j];
        for (size_t i = 0; i < m; ++i) h->ys[(size_t)b * m + i] = E[i] * ys[i] / cc;
        mpcqp_info *inf = &h->info[b];
        inf->status = (tol > 0.0 && !settled) ? MPCQP_MAX_ITER_REACHED : MPCQP_SOLVED; inf->iter = done; inf->rho_updates = 0; inf->reserved = 0; inf->obj_val = 0.0; inf->pri_res = 0.0; inf->dua_res = 0.0; inf->rho = rho;
        if (res) {
            /* the four KKT norms of the EQUALITY-constrained problem as the device reports them (k_eq_solve): |P x + q + A_e' y|, max(|P x|, |A_e' y|, |q|),
             * |A_e x - b|, max(|A_e x|, |b|) over the dynamics rows (the first n_x), on the unscaled solution, from the instance's own P and A */
            double *r = res + (size_t)b * 5; r[0] = 0.0; r[1] = 1.0; r[2] = 0.0; r[3] = 1.0; r[4] = (double)done;
            if (!h->generic && h->Pp) {
                const int64_t ne = h->d.rs;                               /* dynamics rows: 0 .. n_x - 1 */
                double *Pv = (double *)malloc(sizeof(double) * (size_t)(h->nnzP ? h->nnzP : 1)), *Av = (double *)malloc(sizeof(double) * (size_t)(h->nnzA ? h->nnzA : 1));
                double *px = dcalloc(n), *aty = dcalloc(n), *ax = dcalloc(m);
                const double *x = h->xs + (size_t)b * n, *y = h->ys + (size_t)b * m, *q = h->q + (size_t)b * n, *lo = h->l + (size_t)b * m;
                build_values(h, b, Pv, Av);
                for (size_t c = 0; c < n; ++c) {
                    for (int64_t k = h->Pp[c]; k < h->Pp[c + 1]; ++k) {      /* upper triangle: both halves */
                        const int64_t i = h->Pi[k];
                        px[i] += Pv[k] * x[c]; if (i != (int64_t)c) px[c] += Pv[k] * x[i];
                    }
                    for (int64_t k = h->Ap[c]; k < h->Ap[c + 1]; ++k) {
                        const int64_t i = h->Ai[k];
                        if (i < ne) { ax[i] += Av[k] * x[c]; aty[c] += Av[k] * y[i]; }
                    }
                }
                double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
                for (size_t j = 0; j < n; ++j) { n0 = fmax(n0, fabs(px[j] + q[j] + aty[j])); n1 = fmax(n1, fmax(fabs(px[j]), fmax(fabs(aty[j]), fabs(q[j])))); }
                for (int64_t i = 0; i < ne; ++i) { n2 = fmax(n2, fabs(ax[i] - lo[i])); n3 = fmax(n3, fmax(fabs(ax[i]), fabs(lo[i]))); }
                r[0] = n0; r[1] = n1; r[2] = n2; r[3] = n3;
                free(Pv); free(Av); free(px); free(aty); free(ax);
            }
        }
    }
    free(D); free(E); free(xs); free(zs); free(ys); free(xp);
    return MPCQP_OK;
}
/* (the CPU twin keeps one factor per instance: nothing is shared, nothing changes) */
int mpcqp_share_factor(mpcqp_handle *h, int *nshared) { if (nshared) *nshared = 0; return h && h->is_setup ? MPCQP_OK : fail(MPCQP_ERR_STATE, "mpcqp_share_factor before mpcqp_setup"); }
int mpcqp_refactor(mpcqp_handle *h) { return h && h->is_setup ? MPCQP_OK : fail(MPCQP_ERR_STATE, "mpcqp_refactor before mpcqp_setup"); }

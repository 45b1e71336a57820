/*
 * dzg_oracle.c -- CPU restatement of the matteosantama/dantzig hot path.
 * TEST INFRASTRUCTURE ONLY (see dzg_oracle.h).  Plain C, one rounding per
 * floating-point operation, loop order of the reference.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math
 */
#include "dzg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* src/linalg.rs:88-128  Matrix::factorize                                   */
/* ------------------------------------------------------------------------ */
void ora_lu_factorize(double *a, int64_t n, int64_t *p)
{
    for (int64_t k = 0; k + 1 < n; ++k) {
        /* :98-105 first row i >= k maximising |a(i,k)|, strict '>' */
        int64_t mu = k;
        double magnitude = fabs(a[k * n + k]);
        for (int64_t i = k + 1; i < n; ++i) {
            if (fabs(a[i * n + k]) > magnitude) {
                mu = i;
                magnitude = fabs(a[i * n + k]);
            }
        }
        /* :107-113 swap rows k and mu for columns j >= k only */
        for (int64_t j = k; j < n; ++j) {
            double t = a[mu * n + j];
            a[mu * n + j] = a[k * n + j];
            a[k * n + j] = t;
        }
        p[k] = mu; /* :114 */

        /* :116-125 zero pivot silently skipped */
        double pivot = a[k * n + k];
        if (pivot != 0.0) {
            for (int64_t i = k + 1; i < n; ++i) {
                a[i * n + k] /= pivot;
                const double lik = a[i * n + k];
                for (int64_t j = k + 1; j < n; ++j) {
                    double adjustment = lik * a[k * n + j];
                    a[i * n + j] -= adjustment;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* src/linalg.rs:282-299  LU::solve                                          */
/* ------------------------------------------------------------------------ */
void ora_lu_solve(const double *lu, int64_t n, const int64_t *p, double *b)
{
    /* :286-291 forward, swaps interleaved */
    for (int64_t k = 0; k + 1 < n; ++k) {
        double t = b[k];
        b[k] = b[p[k]];
        b[p[k]] = t;
        for (int64_t i = k + 1; i < n; ++i) {
            double prod = b[k] * lu[i * n + k];
            b[i] -= prod;
        }
    }
    /* :292-297 backward, j ascending inside each row */
    for (int64_t i = n - 1; i >= 0; --i) {
        for (int64_t j = i + 1; j < n; ++j) {
            double prod = lu[i * n + j] * b[j];
            b[i] -= prod;
        }
        b[i] /= lu[i * n + i];
    }
}

/* src/linalg.rs:8-10 */
void ora_lu_solve_full(double *a, int64_t n, double *b)
{
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 1 ? n - 1 : 1));
    ora_lu_factorize(a, n, p);
    ora_lu_solve(a, n, p, b);
    free(p);
}

/* src/linalg.rs:40-48 */
void ora_matrix_t(const double *in, int64_t nrows, int64_t ncols, double *out)
{
    for (int64_t j = 0; j < ncols; ++j)
        for (int64_t i = 0; i < nrows; ++i)
            out[j * nrows + i] = in[i * ncols + j];
}

/* ------------------------------------------------------------------------ */
/* src/linalg.rs:254-270  From<&Matrix> for CscMatrix                        */
/* ------------------------------------------------------------------------ */
int64_t ora_csc_from_dense(const double *dense, int64_t nrows, int64_t ncols,
                           int64_t *col_ptr, int64_t *row_idx, double *val)
{
    int64_t nnz = 0;
    col_ptr[0] = 0;
    for (int64_t j = 0; j < ncols; ++j) {
        for (int64_t i = 0; i < nrows; ++i) {
            double v = dense[i * ncols + j];
            if (v != 0.0) {
                row_idx[nnz] = i;
                val[nnz] = v;
                ++nnz;
            }
        }
        col_ptr[j + 1] = nnz;
    }
    return nnz;
}

/* src/linalg.rs:180-186 */
void ora_csc_column(int64_t nrows, const int64_t *col_ptr, const int64_t *row_idx,
                    const double *val, int64_t j, double *out)
{
    for (int64_t i = 0; i < nrows; ++i) out[i] = 0.0;
    for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e) out[row_idx[e]] = val[e];
}

/* src/linalg.rs:131-140 */
void ora_csc_to_dense(int64_t nrows, int64_t ncols, const int64_t *col_ptr,
                      const int64_t *row_idx, const double *val, double *out)
{
    for (int64_t i = 0; i < nrows * ncols; ++i) out[i] = 0.0;
    for (int64_t j = 0; j < ncols; ++j)
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e)
            out[row_idx[e] * ncols + j] = val[e];
}

/* src/linalg.rs:188-192 + 199-207.  collect_columns densifies then drops the
 * zeros again, so the stored entries of the gathered matrix are exactly the
 * stored entries of the source columns: iterate those directly. */
void ora_csc_neg_t_dot(const int64_t *col_ptr, const int64_t *row_idx,
                       const double *val, const int64_t *cols, int64_t ncols_sel,
                       const double *v, double *out)
{
    for (int64_t k = 0; k < ncols_sel; ++k) {
        int64_t j = cols[k];
        double acc = 0.0; /* Iterator::sum identity, SURVEY App. A.7 */
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e) {
            double prod = val[e] * -v[row_idx[e]];
            acc = acc + prod;
        }
        out[k] = acc;
    }
}

/* ------------------------------------------------------------------------ */
/* src/simplex.rs:423-437  find_first_pivot                                  */
/* ------------------------------------------------------------------------ */
int64_t ora_find_first_pivot(const double *y, const double *ybar, int64_t len)
{
    int64_t best = -1;
    double best_ratio = 0.0;
    for (int64_t k = 0; k < len; ++k) {
        if (!(ybar[k] > 0.0)) continue;
        double ratio = -y[k] / ybar[k];
        if (best < 0) {
            best = k;
            best_ratio = ratio;
        } else if (ratio > best_ratio) {
            best = k;
            best_ratio = ratio;
        }
    }
    return best;
}

/* src/simplex.rs:439-461  find_second_pivot */
int64_t ora_find_second_pivot(double mu, const double *y, const double *ybar,
                              const double *dy, int64_t len)
{
    int64_t best = -1;
    double best_ratio = 0.0;
    for (int64_t k = 0; k < len; ++k) {
        double scaled = mu * ybar[k];
        double denominator = y[k] + scaled;
        double ratio = dy[k] / denominator;
        if (!(ratio > 0.0)) continue;
        if (best < 0) {
            best = k;
            best_ratio = ratio;
        } else if (ratio > best_ratio) {
            best = k;
            best_ratio = ratio;
        }
    }
    return best;
}

/* src/simplex.rs:464-468 safe_divide; returns 0 on the assert path */
static int safe_divide(double x, double y, double *out)
{
    double div = (x == 0.0 && y == 0.0) ? 0.0 : x / y;
    *out = div;
    return !(isinf(div) || isnan(div));
}

/* src/simplex.rs:410-421 pivot (free function) */
static void pivot_vec(double *data, const double *delta, int64_t len, int64_t index,
                      double step_length)
{
    for (int64_t i = 0; i < len; ++i) {
        if (i == index) {
            data[i] = step_length;
        } else {
            double prod = step_length * delta[i];
            data[i] -= prod;
        }
    }
}

/* The factorisation the iteration calls.  The default is the literal restatement above; the
 * twin library libdzg_oracle_blocked.so is this file compiled with
 * -DORA_LU=ora_lu_factorize_blocked (dzg_oracle_blocked.c: the same operations on every element
 * in the same order, bit-equal factors, rearranged for the cache and for several cores) and only
 * writes pivot-log fixtures at sizes where the literal loop needs minutes per pivot. */
#ifndef ORA_LU
#define ORA_LU ora_lu_factorize
#else
void ORA_LU(double *a, int64_t n, int64_t *p);
#endif

/* Workspace for one solve */
typedef struct {
    double *bm;    /* m*m  basis matrix, row-major: B[r][c] = A[r][basis[c]] */
    double *bt;    /* m*m  its transpose                                     */
    int64_t *p;    /* m    LU pivots                                         */
    double *dx, *dz, *v;
} ora_work;

/* Input format for fixtures at sizes where a CSC with 64-bit indices does not fit the build
 * container (32768 x 65536: 34 GB): row_idx == NULL means "val is the structural block, dense
 * column-major m x (n - m); column j >= n - m is the unit column of row j - (n - m)" -- the
 * benchmark convention of SURVEY 8(d).  The stored entries of a column are then its nonzeros in
 * ascending row order, exactly what the reference's CSC holds (src/linalg.rs:254-270 drops exact
 * zeros), so every loop below visits the same entries in the same order in both formats
 * (tests/test_oracle_kats.py holds the two bit-equal). */
static int dense_input(const ora_simplex *s) { return s->row_idx == NULL; }

/* src/simplex.rs:270-272 basis_matrix + linalg.rs:236-238 to_dense */
static void gather_basis(const ora_simplex *s, double *bm)
{
    const int64_t m = s->m;
    for (int64_t i = 0; i < m * m; ++i) bm[i] = 0.0;
    if (dense_input(s)) {
        const int64_t ns = s->n - m;
        for (int64_t c = 0; c < m; ++c) {
            int64_t j = s->basis[c];
            if (j >= ns) {
                bm[(j - ns) * m + c] = 1.0;
                continue;
            }
            const double *col = s->val + j * m;
            for (int64_t r = 0; r < m; ++r)
                if (col[r] != 0.0) bm[r * m + c] = col[r];
        }
        return;
    }
    for (int64_t c = 0; c < m; ++c) {
        int64_t j = s->basis[c];
        for (int64_t e = s->col_ptr[j]; e < s->col_ptr[j + 1]; ++e)
            bm[s->row_idx[e] * m + c] = s->val[e];
    }
}

/* CscMatrix::column (src/linalg.rs:180-186) in either input format */
static void column_of(const ora_simplex *s, int64_t j, double *out)
{
    if (!dense_input(s)) {
        ora_csc_column(s->m, s->col_ptr, s->row_idx, s->val, j, out);
        return;
    }
    const int64_t m = s->m, ns = s->n - m;
    for (int64_t i = 0; i < m; ++i) out[i] = 0.0;
    if (j >= ns) {
        out[j - ns] = 1.0;
        return;
    }
    const double *col = s->val + j * m;
    for (int64_t r = 0; r < m; ++r)
        if (col[r] != 0.0) out[r] = col[r];
}

/* collect_columns(n).neg_t_dot(v) (src/linalg.rs:188-207) in either input format */
static void price(const ora_simplex *s, const double *v, double *out)
{
    const int64_t m = s->m, q = s->n - m;
    if (!dense_input(s)) {
        ora_csc_neg_t_dot(s->col_ptr, s->row_idx, s->val, s->nonbasis, q, v, out);
        return;
    }
    for (int64_t k = 0; k < q; ++k) {
        int64_t j = s->nonbasis[k];
        double acc = 0.0;
        if (j >= q) { /* ns == q: the unit column's one stored entry */
            double prod = 1.0 * -v[j - q];
            acc = acc + prod;
        } else {
            const double *col = s->val + j * m;
            for (int64_t r = 0; r < m; ++r) {
                if (col[r] == 0.0) continue; /* not a stored entry */
                double prod = col[r] * -v[r];
                acc = acc + prod;
            }
        }
        out[k] = acc;
    }
}

/* src/simplex.rs:226-229 solve_for_dx */
static void solve_for_dx(const ora_simplex *s, ora_work *w, int64_t j)
{
    gather_basis(s, w->bm); /* basis_matrix.clone().to_dense() */
    column_of(s, j, w->dx);
    ORA_LU(w->bm, s->m, w->p);
    ora_lu_solve(w->bm, s->m, w->p, w->dx);
}

/* src/simplex.rs:231-236 solve_for_dz; pos = b_key[i] */
static void solve_for_dz(const ora_simplex *s, ora_work *w, int64_t pos)
{
    const int64_t m = s->m;
    gather_basis(s, w->bm);
    ora_matrix_t(w->bm, m, m, w->bt); /* .to_dense().t() */
    for (int64_t i = 0; i < m; ++i) w->v[i] = 0.0;
    w->v[pos] = 1.0;
    ORA_LU(w->bt, m, w->p); /* a second, independent LU (App. A.4) */
    ora_lu_solve(w->bt, m, w->p, w->v);
    price(s, w->v, w->dz);
}

/* src/simplex.rs:253-268 Simplex::pivot + :239-251 swap */
static int do_pivot(ora_simplex *s, ora_work *w, int64_t b_i, int64_t n_j)
{
    const int64_t m = s->m, q = s->n - s->m;
    double t, sd, t_bar, s_bar;
    int ok = 1;
    ok &= safe_divide(s->x[b_i], w->dx[b_i], &t);
    ok &= safe_divide(s->z[n_j], w->dz[n_j], &sd);
    ok &= safe_divide(s->xbar[b_i], w->dx[b_i], &t_bar);
    ok &= safe_divide(s->zbar[n_j], w->dz[n_j], &s_bar);
    if (!ok) return 0;
    pivot_vec(s->x, w->dx, m, b_i, t);
    pivot_vec(s->xbar, w->dx, m, b_i, t_bar);
    pivot_vec(s->z, w->dz, q, n_j, sd);
    pivot_vec(s->zbar, w->dz, q, n_j, s_bar);
    /* swap: entering variable takes the leaving variable's slot and vice versa */
    int64_t i = s->basis[b_i], j = s->nonbasis[n_j];
    s->basis[b_i] = j;
    s->nonbasis[n_j] = i;
    return 1;
}

#define ORA_EPSILON 1e-12 /* src/simplex.rs:9 */

int ora_simplex_solve(ora_simplex *s, int64_t max_iter, int64_t *iterations,
                      ora_pivot *log, int64_t log_cap)
{
    const int64_t m = s->m, q = s->n - s->m;
    ora_work w;
    size_t mm = (size_t)(m > 0 ? m : 1);
    w.bm = (double *)malloc(sizeof(double) * mm * mm);
    w.bt = (double *)malloc(sizeof(double) * mm * mm);
    w.p = (int64_t *)malloc(sizeof(int64_t) * mm);
    w.dx = (double *)malloc(sizeof(double) * mm);
    w.v = (double *)malloc(sizeof(double) * mm);
    w.dz = (double *)malloc(sizeof(double) * (size_t)(q > 0 ? q : 1));

    int status = ORA_ITER_LIMIT;
    int64_t it = 0;
    for (;;) {
        /* ---- status, src/simplex.rs:274-306 ---- */
        int64_t pj = ora_find_first_pivot(s->z, s->zbar, q);
        int64_t pi = ora_find_first_pivot(s->x, s->xbar, m);
        int kind;
        double mu;
        if (pj >= 0 && pi >= 0) {
            double primal = -s->x[pi] / s->xbar[pi];
            double dual = -s->z[pj] / s->zbar[pj];
            if (primal <= ORA_EPSILON && dual <= ORA_EPSILON) {
                status = ORA_OPTIMAL;
                break;
            }
            if (primal < dual) {
                kind = ORA_STEP_PRIMAL;
                mu = dual;
            } else {
                kind = ORA_STEP_DUAL;
                mu = primal;
            }
        } else if (pj >= 0) {
            kind = ORA_STEP_PRIMAL;
            mu = -s->z[pj] / s->zbar[pj];
        } else if (pi >= 0) {
            kind = ORA_STEP_DUAL;
            mu = -s->x[pi] / s->xbar[pi];
        } else {
            status = ORA_PANIC; /* :304 */
            break;
        }
        if (it >= max_iter) {
            status = ORA_ITER_LIMIT;
            break;
        }
        if (m == 0) { /* n - 1 underflow in factorize/solve: reference panic */
            status = ORA_PANIC;
            break;
        }

        int64_t b_i, n_j;
        if (kind == ORA_STEP_PRIMAL) { /* :308-318 */
            n_j = pj;
            solve_for_dx(s, &w, s->nonbasis[n_j]);
            b_i = ora_find_second_pivot(mu, s->x, s->xbar, w.dx, m);
            if (b_i < 0) {
                status = ORA_UNBOUNDED;
                break;
            }
            solve_for_dz(s, &w, b_i);
        } else { /* :320-330 */
            b_i = pi;
            solve_for_dz(s, &w, b_i);
            n_j = ora_find_second_pivot(mu, s->z, s->zbar, w.dz, q);
            if (n_j < 0) {
                status = ORA_INFEASIBLE;
                break;
            }
            solve_for_dx(s, &w, s->nonbasis[n_j]);
        }
        if (log && it < log_cap) {
            log[it].kind = kind;
            log[it].entering = s->nonbasis[n_j];
            log[it].leaving = s->basis[b_i];
            log[it].mu = mu;
        }
        if (!do_pivot(s, &w, b_i, n_j)) {
            status = ORA_PANIC; /* safe_divide assert, :466: the pivot was chosen, not executed */
            break;
        }
        ++it;
    }
    if (iterations) *iterations = it;
    free(w.bm);
    free(w.bt);
    free(w.p);
    free(w.dx);
    free(w.v);
    free(w.dz);
    return status;
}

/* src/simplex.rs:345-352 */
double ora_objective_value(const ora_simplex *s)
{
    double sum = 0.0;
    for (int64_t pos = 0; pos < s->m; ++pos) {
        double prod = s->c[s->basis[pos]] * s->x[pos];
        sum = sum + prod;
    }
    return s->constant + sum;
}

/* ------------------------------------------------------------------------ */
/* src/simplex.rs:123-224  Simplex::new                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t *var;  /* internal ids */
    double *coef;
    int64_t len, cap;
    double b;
} row_t;

static void row_push(row_t *r, int64_t var, double coef)
{
    if (r->len == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 8;
        r->var = (int64_t *)realloc(r->var, sizeof(int64_t) * (size_t)r->cap);
        r->coef = (double *)realloc(r->coef, sizeof(double) * (size_t)r->cap);
    }
    r->var[r->len] = var;
    r->coef[r->len] = coef;
    r->len++;
}

int ora_build_standard_form(const ora_model *md, ora_stdform *out)
{
    const int64_t V = md->nvars;
    memset(out, 0, sizeof(*out));
    int64_t *ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));
    for (int64_t u = 0; u < V; ++u) ord[u] = -1;
    int64_t nseen = 0;

    /* :124-151 bound rows, in order of first appearance; ub row before lb row */
    int64_t extra_cap = 2 * V + 1, nextra = 0;
    row_t *extra = (row_t *)calloc((size_t)extra_cap, sizeof(row_t));
    int64_t total_terms = md->obj_nterms + (md->ncons ? md->con_ptr[md->ncons] : 0);
    for (int64_t t = 0; t < total_terms; ++t) {
        int64_t u = t < md->obj_nterms ? md->obj_var[t] : md->con_var[t - md->obj_nterms];
        if (ord[u] >= 0) continue;
        ord[u] = nseen++;
        int64_t pos = 2 * ord[u], neg = 2 * ord[u] + 1;
        if (md->has_ub[u]) {
            row_t *r = &extra[nextra++];
            row_push(r, pos, 1.0);
            row_push(r, neg, -1.0);
            r->b = md->ub[u];
        }
        if (md->has_lb[u]) {
            row_t *r = &extra[nextra++];
            row_push(r, pos, -1.0);
            row_push(r, neg, 1.0);
            r->b = -md->lb[u];
        }
    }

    /* :153-166 split every expression, append the bound rows, inject slacks */
    const int64_t m = md->ncons + nextra;
    const int64_t slack0 = 2 * nseen; /* internal id of row r's slack = slack0 + r */
    row_t *rows = (row_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(row_t));
    for (int64_t r = 0; r < md->ncons; ++r) {
        for (int64_t e = md->con_ptr[r]; e < md->con_ptr[r + 1]; ++e) {
            int64_t u = md->con_var[e];
            row_push(&rows[r], 2 * ord[u], md->con_coef[e]);      /* src/model.rs:11-22 */
            row_push(&rows[r], 2 * ord[u] + 1, -md->con_coef[e]);
        }
        rows[r].b = md->con_b[r];
    }
    for (int64_t r = 0; r < nextra; ++r) rows[md->ncons + r] = extra[r];
    for (int64_t r = 0; r < m; ++r) row_push(&rows[r], slack0 + r, 1.0); /* :19-31 */

    /* :168-176 column index = order of first appearance */
    const int64_t n = 2 * nseen + m;
    int64_t *index_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *id_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) index_of[i] = -1;
    int64_t next = 0;
    for (int64_t t = 0; t < md->obj_nterms; ++t) {
        int64_t u = md->obj_var[t];
        for (int s = 0; s < 2; ++s) {
            int64_t id = 2 * ord[u] + s;
            if (index_of[id] < 0) {
                index_of[id] = next;
                id_of[next++] = id;
            }
        }
    }
    for (int64_t r = 0; r < m; ++r)
        for (int64_t e = 0; e < rows[r].len; ++e) {
            int64_t id = rows[r].var[e];
            if (index_of[id] < 0) {
                index_of[id] = next;
                id_of[next++] = id;
            }
        }

    /* :39-49 Objective::new, coefficients scattered by ASSIGNMENT */
    out->m = m;
    out->n = n;
    out->c = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (int64_t t = 0; t < md->obj_nterms; ++t) {
        int64_t u = md->obj_var[t];
        out->c[index_of[2 * ord[u]]] = md->obj_coef[t];
        out->c[index_of[2 * ord[u] + 1]] = -md->obj_coef[t];
    }
    out->constant = md->obj_const;

    /* :190-201 initial basis = slacks, x = rhs, z = -c */
    out->basis = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    out->x = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    out->nonbasis = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n - m > 0 ? n - m : 1));
    out->z = (double *)malloc(sizeof(double) * (size_t)(n - m > 0 ? n - m : 1));
    int64_t nb = 0, nn = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t id = id_of[i];
        if (id >= slack0) {
            out->basis[nb] = i;
            out->x[nb] = rows[id - slack0].b;
            nb++;
        } else {
            out->nonbasis[nn] = i;
            out->z[nn] = -out->c[i];
            nn++;
        }
    }

    /* :62-81 sparsify: coords -> dense (assignment) -> CSC without zeros */
    double *dense = (double *)calloc((size_t)(m * n > 0 ? m * n : 1), sizeof(double));
    for (int64_t r = 0; r < m; ++r)
        for (int64_t e = 0; e < rows[r].len; ++e)
            dense[r * n + index_of[rows[r].var[e]]] = rows[r].coef[e];
    out->col_ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    out->row_idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m * n > 0 ? m * n : 1));
    out->val = (double *)malloc(sizeof(double) * (size_t)(m * n > 0 ? m * n : 1));
    out->nnz = ora_csc_from_dense(dense, m, n, out->col_ptr, out->row_idx, out->val);
    free(dense);

    out->pos_col = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));
    out->neg_col = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));
    for (int64_t u = 0; u < V; ++u) {
        out->pos_col[u] = ord[u] >= 0 ? index_of[2 * ord[u]] : -1;
        out->neg_col[u] = ord[u] >= 0 ? index_of[2 * ord[u] + 1] : -1;
    }

    for (int64_t r = 0; r < m; ++r) {
        free(rows[r].var);
        free(rows[r].coef);
    }
    free(rows);
    free(extra);
    free(ord);
    free(index_of);
    free(id_of);
    return 0;
}

void ora_stdform_free(ora_stdform *sf)
{
    free(sf->col_ptr);
    free(sf->row_idx);
    free(sf->val);
    free(sf->c);
    free(sf->basis);
    free(sf->nonbasis);
    free(sf->x);
    free(sf->z);
    free(sf->pos_col);
    free(sf->neg_col);
    memset(sf, 0, sizeof(*sf));
}

/* src/simplex.rs:354-371 */
void ora_solution(const ora_simplex *s, const ora_stdform *sf, int64_t nvars, double *value)
{
    int64_t *pos_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->n > 0 ? s->n : 1));
    for (int64_t i = 0; i < s->n; ++i) pos_of[i] = -1;
    for (int64_t k = 0; k < s->m; ++k) pos_of[s->basis[k]] = k;
    for (int64_t u = 0; u < nvars; ++u) {
        if (sf->pos_col[u] < 0) {
            value[u] = 0.0; /* src/pyobjs.rs:163-165 unknown variable -> 0.0 */
            continue;
        }
        double pos = pos_of[sf->pos_col[u]] >= 0 ? s->x[pos_of[sf->pos_col[u]]] : 0.0;
        double neg = pos_of[sf->neg_col[u]] >= 0 ? s->x[pos_of[sf->neg_col[u]]] : 0.0;
        value[u] = pos - neg;
    }
    free(pos_of);
}

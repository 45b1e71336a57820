/*
 * dzg_oracle.h -- CPU restatement of the matteosantama/dantzig simplex hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline.  The product
 * (dantzig_amd/) never links, imports or calls it.
 *
 * Parity pin: the Rust reference cannot be compiled here (no cargo/rustc, see
 * DESIGN.md) so this restatement is pinned by the reference's own known-answer
 * tests, committed as data under tests/golden/reference_kats.json:
 *   src/linalg.rs:306-446  (LU, solve, CSC, transpose, neg_t_dot KATs, exact)
 *   src/simplex.rs:484-796 (16 solver KATs, 1e-12)
 *   tests/test_optimize.py, tests/test_exceptions.py (Python KATs, exact ==)
 *
 * All arithmetic is IEEE-754 binary64, one rounding per operation, in the
 * loop order of the reference (compile with -ffp-contract=off, no fast-math).
 * Every function cites the reference file:line it follows.
 */
#ifndef DZG_ORACLE_H
#define DZG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with include/dantzig_amd.h */
enum {
    ORA_OPTIMAL = 0,
    ORA_UNBOUNDED = 1,   /* src/simplex.rs:313 Error::Unbounded  */
    ORA_INFEASIBLE = 2,  /* src/simplex.rs:325 Error::Infeasible */
    ORA_ITER_LIMIT = 3,  /* not in the reference (it recurses without a cap) */
    ORA_PANIC = 5        /* a reference panic path: safe_divide assert (:466),
                            "unexpected code path" (:304), n == 0 underflow   */
};

enum { ORA_STEP_PRIMAL = 0, ORA_STEP_DUAL = 1 };

/* ---- src/linalg.rs: dense row-major Matrix + LU -------------------------- */

/* Matrix::factorize, src/linalg.rs:88-128.  a: n*n row-major, in place.
 * p: n-1 pivot rows.  LINPACK convention: swaps touch columns k..n only. */
void ora_lu_factorize(double *a, int64_t n, int64_t *p);

/* LU::solve, src/linalg.rs:282-299.  b: n, in place. */
void ora_lu_solve(const double *lu, int64_t n, const int64_t *p, double *b);

/* lu_solve, src/linalg.rs:8-10 (factorize then solve; a is destroyed). */
void ora_lu_solve_full(double *a, int64_t n, double *b);

/* Matrix::t, src/linalg.rs:40-48.  in: nrows*ncols row-major -> out: ncols*nrows */
void ora_matrix_t(const double *in, int64_t nrows, int64_t ncols, double *out);

/* ---- src/linalg.rs: CSC --------------------------------------------------- */

/* From<&Matrix> for CscMatrix, src/linalg.rs:254-270.  Exact zeros dropped.
 * col_ptr: ncols+1; row_idx/val: capacity nrows*ncols.  Returns nnz. */
int64_t ora_csc_from_dense(const double *dense, int64_t nrows, int64_t ncols,
                           int64_t *col_ptr, int64_t *row_idx, double *val);

/* CscMatrix::column, src/linalg.rs:180-186. out: nrows */
void ora_csc_column(int64_t nrows, const int64_t *col_ptr, const int64_t *row_idx,
                    const double *val, int64_t j, double *out);

/* Matrix::from(&CscMatrix), src/linalg.rs:131-140. out: nrows*ncols row-major */
void ora_csc_to_dense(int64_t nrows, int64_t ncols, const int64_t *col_ptr,
                      const int64_t *row_idx, const double *val, double *out);

/* CscMatrix::collect_columns(cols).neg_t_dot(v), src/linalg.rs:188-207:
 * out[k] = sum over stored entries of column cols[k], ascending row, of
 * val * (-v[row]), accumulated left to right from 0.0. */
void ora_csc_neg_t_dot(const int64_t *col_ptr, const int64_t *row_idx,
                       const double *val, const int64_t *cols, int64_t ncols_sel,
                       const double *v, double *out);

/* ---- src/simplex.rs: the iteration ---------------------------------------- */

/* find_first_pivot, src/simplex.rs:423-437.  Returns POSITION k (the reference
 * returns index_lookup[k]; callers map back through b_key/n_key) or -1. */
int64_t ora_find_first_pivot(const double *y, const double *ybar, int64_t len);

/* find_second_pivot, src/simplex.rs:439-461.  Returns position or -1. */
int64_t ora_find_second_pivot(double mu, const double *y, const double *ybar,
                              const double *dy, int64_t len);

/* State of `Simplex` after Simplex::new (src/simplex.rs:84-112,209-223). */
typedef struct {
    int64_t m, n;            /* rows, all columns incl. slacks              */
    const int64_t *col_ptr;  /* n+1                                          */
    const int64_t *row_idx;  /* nnz, ascending within a column; NULL: val is the
                                structural block dense column-major m x (n-m), columns
                                n-m..n-1 are the unit columns of rows 0..m-1 (fixtures at
                                sizes whose 64-bit CSC does not fit the build container;
                                same entries visited in the same order, exact zeros skipped) */
    const double *val;       /* nnz, no explicit zeros                       */
    const double *c;         /* n objective coefficients (core MAXIMISES)    */
    double constant;
    int64_t *basis;          /* m  variable index per basic position  (b)    */
    int64_t *nonbasis;       /* n-m                                   (n)    */
    double *x, *xbar;        /* m                                            */
    double *z, *zbar;        /* n-m                                          */
} ora_simplex;

typedef struct {
    int32_t kind;      /* ORA_STEP_PRIMAL / ORA_STEP_DUAL */
    int64_t entering;  /* variable index j                */
    int64_t leaving;   /* variable index i                */
    double mu;
} ora_pivot;

/* Simplex::solve, src/simplex.rs:332-343 (iterative, same decisions).
 * log may be NULL; at most log_cap entries are written.  Returns status. */
int ora_simplex_solve(ora_simplex *s, int64_t max_iter, int64_t *iterations,
                      ora_pivot *log, int64_t log_cap);

/* objective_value, src/simplex.rs:345-352 (summed in basis-position order;
 * the reference sums in HashMap order). */
double ora_objective_value(const ora_simplex *s);

/* ---- src/simplex.rs:123-224 Simplex::new: the standard-form builder ------- */

typedef struct {
    int64_t nvars;                 /* user variables in the table              */
    const int32_t *has_lb, *has_ub;
    const double *lb, *ub;
    int64_t obj_nterms;            /* objective terms, in expression order     */
    const int64_t *obj_var;        /* index into the variable table            */
    const double *obj_coef;
    double obj_const;
    int64_t ncons;                 /* inequality rows  coef.x <= b             */
    const int64_t *con_ptr;        /* ncons+1                                  */
    const int64_t *con_var;
    const double *con_coef;
    const double *con_b;
} ora_model;

typedef struct {
    int64_t m, n, nnz;
    int64_t *col_ptr, *row_idx;
    double *val, *c;
    double constant;
    int64_t *basis, *nonbasis;
    double *x, *z;
    int64_t *pos_col, *neg_col;    /* nvars: column of x+ / x- ; -1 if unseen  */
} ora_stdform;

/* Allocates every array in *out with malloc; release with ora_stdform_free. */
int ora_build_standard_form(const ora_model *model, ora_stdform *out);
void ora_stdform_free(ora_stdform *sf);

/* Simplex::solution, src/simplex.rs:354-371: value[u] = x[pos]-x[neg]. */
void ora_solution(const ora_simplex *s, const ora_stdform *sf, int64_t nvars,
                  double *value);

#ifdef __cplusplus
}
#endif
#endif

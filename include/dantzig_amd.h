/*
 * dantzig_amd.h -- C ABI of the MI355X-native parametric self-dual simplex core.
 *
 * This is the drop-in boundary for the hot path of matteosantama/dantzig
 * (src/simplex.rs + src/linalg.rs).  The reference has no C ABI: its seam is the
 * PyO3 function `dantzig.rust.solve` (src/lib.rs:16-27) and, inside Rust, the pair
 * `Simplex::new(..)` / `Simplex::solve()` (src/simplex.rs:123, :332).  A Rust host
 * binds this header with a plain `extern "C"` block (INTEGRATION.md shows the stub);
 * the Python package binds it with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-allocated host memory
 *     unless a parameter says "device"; the library never frees caller memory;
 *   - no exceptions or unwinding cross this boundary: every entry point returns a
 *     dzg_status (>= 0: solver outcome, < 0: call failed);
 *   - one solve per handle at a time (thread-compatible, not thread-safe);
 *   - all arithmetic is IEEE-754 binary64; indices are int64 at the boundary.
 *
 * Level 1 (dzg_solver_*, dzg_core_solve) takes the state `Simplex::new` leaves
 * behind and replaces `Simplex::solve`.  Level 2 (dzg_model_solve) also replaces
 * `Simplex::new` and `PySolution::from`, i.e. the whole of `dantzig.rust.solve`.
 * The dzg_kernel_* entry points expose single reference functions for parity tests.
 */
#ifndef DANTZIG_AMD_H
#define DANTZIG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZG_ABI_VERSION 4

/* Outcome codes.  0..2 mirror the reference: Ok / Error::Unbounded / Error::Infeasible
 * (src/error.rs:3-7, src/simplex.rs:313,325).  The rest do not exist in the
 * reference, which recurses without a cap and panics instead. */
typedef enum {
    DZG_OPTIMAL = 0,
    DZG_UNBOUNDED = 1,
    DZG_INFEASIBLE = 2,
    DZG_ITER_LIMIT = 3,   /* stopped by max_iter / the run budget, state is resumable */
    DZG_SINGULAR = 4,     /* fast numerics: basis inverse lost accuracy                */
    DZG_PANIC = 5,        /* a reference panic path: safe_divide assert (src/simplex.rs:466),
                             "unexpected code path" (:304), zero-row LP (n-1 underflow) */
    DZG_RUNNING = 6,      /* internal: not terminated yet                              */
    DZG_NEAR_TIE = 7,     /* fast numerics, opts.near_tie_action = STOP: stopped BEFORE a pivot
                             whose choice (status(), find_first_pivot, find_second_pivot,
                             src/simplex.rs:274-306,423-461) is within rounding of a tie, so the
                             reference's arithmetic may choose differently.  The state is that
                             of the last executed pivot; dzg_solver_run resumes (that decision is
                             then taken as FAST sees it and counted in near_ties)         */
    DZG_E_DEVICE = -1,    /* HIP error, or no usable GPU (the product has no CPU path) */
    DZG_E_ARG = -2,       /* malformed input                                           */
    DZG_E_NOMEM = -3
} dzg_status;

typedef enum { DZG_STEP_PRIMAL = 0, DZG_STEP_DUAL = 1 } dzg_step_kind;

/* How dx = B^-1 a_j and v = B^-T e_p are obtained each iteration.
 *  STRICT: exactly the reference's arithmetic -- gather B, dense LU with partial
 *          pivoting of B and, separately, of B^T, forward/back substitution in the
 *          reference's operation order (src/linalg.rs:88-128,282-299), unfused.
 *          Bit-identical to the CPU restatement; O(m^3) per iteration.
 *  FAST:   the basis inverse is kept resident in HBM and updated per pivot; FTRAN is
 *          one GEMV, BTRAN one row read.  Same pivot rule; results agree with STRICT
 *          up to rounding (pivot choices differ only on near-ties).
 *  AUTO:   STRICT when m <= auto_strict_rows, else FAST. */
typedef enum { DZG_NUMERICS_STRICT = 0, DZG_NUMERICS_FAST = 1, DZG_NUMERICS_AUTO = 2 } dzg_numerics;

/* Pricing kernel choice for dz = -N^T v (src/linalg.rs:199-207). */
typedef enum {
    DZG_PRICE_AUTO = 0,
    DZG_PRICE_SEQ = 1,   /* one lane per column through an LDS-transposed tile: sums in the
                            reference's ascending-row order, bit-identical to neg_t_dot     */
    DZG_PRICE_WAVE = 2,  /* one wave per column, lane-strided partial sums + shuffle tree   */
    DZG_PRICE_TREE = 3   /* several columns per wave, register accumulators + shuffle tree: the
                            streaming kernel of FAST numerics; sums depend only on m, so dz is
                            bit-identical across column shardings (AUTO picks it for FAST and
                            DZG_PRICE_SEQ for STRICT)                                         */
} dzg_price_kernel;

/* The LP in the state `Simplex::new` produces (src/simplex.rs:209-223):
 *      maximise  c.x + constant   subject to   [A_struct | slacks] x = rhs,  x >= 0
 * Variables are numbered 0..n-1 in the reference's first-appearance order
 * (src/simplex.rs:168-176).  Structural columns are stored dense column-major; slack
 * columns are unit vectors and are never stored. */
typedef struct {
    int64_t m;               /* rows                                                    */
    int64_t n;               /* all variables, structural + slack                       */
    int64_t n_struct;        /* structural columns held in `a`                          */
    const double *a;         /* column-major m x n_struct, leading dimension lda >= m   */
    int64_t lda;
    const int64_t *var_col;  /* n entries: >= 0 structural column of variable v;
                                < 0: v is the slack of row (-1 - var_col[v]).
                                NULL = benchmark convention: variable v <
                                n_struct is column v, variable n_struct + i is slack i */
    const double *c;         /* n objective coefficients (the core MAXIMISES)           */
    double constant;
    const int64_t *basis;    /* m   initial basic variable per position  (Simplex.b)    */
    const int64_t *nonbasis; /* n-m initial nonbasic variable per position (Simplex.n)  */
    const double *x;         /* m   initial x (= rhs)                                   */
    const double *z;         /* n-m initial z (= -c of the nonbasic variables)          */
    /* Sparse alternative to `a` (used when a == NULL): the structural block in CSC, exactly
     * the reference's CscMatrix (src/linalg.rs:161-168): rows ascending inside a column, no
     * explicit zeros.  Kept as CSC on the device (12 bytes per nonzero). */
    const int64_t *col_ptr;  /* n_struct + 1                                            */
    const int32_t *row_idx;  /* nnz                                                     */
    const double *val;       /* nnz                                                     */
    /* The perturbation vectors of the parametric method (Simplex.x_bar, Simplex.z_bar,
     * src/simplex.rs:84-112).  NULL = all ones, as Simplex::new sets them (src/simplex.rs:219-220).
     * Given together with a non-slack `basis`, its `x` and `z`, they RESUME a solve from the state
     * a dzg_result handed back (basis, nonbasis, x, xbar, z, zbar): the basis is factorised on the
     * device and the loop goes on where it stopped. */
    const double *xbar;      /* m    or NULL                                             */
    const double *zbar;      /* n-m  or NULL                                             */
} dzg_lp;

typedef struct {
    int32_t numerics;         /* dzg_numerics, default AUTO                             */
    int32_t price_kernel;     /* dzg_price_kernel, default AUTO                         */
    int32_t device;           /* HIP device ordinal, default 0                          */
    int32_t auto_strict_rows; /* AUTO threshold, default 192                            */
    int64_t max_iter;         /* default 10,000,000                                     */
    double epsilon;           /* optimality tolerance, default 1e-12 (src/simplex.rs:9) */
    int64_t log_capacity;     /* pivots kept in the device log, default min(max_iter, 2^22) */
    int32_t poll_interval;    /* FAST: iterations enqueued between host status polls, default 32 */
    int32_t profile;          /* bit mask, bit (1 << DZG_K_*) set: time that kernel class with
                                 HIP events (slower); 0 = no timing, -1 = every class.
                                 Bits 16..23: a sampling stride S > 1 -- only every S-th
                                 iteration of a batch is stamped (an event pair costs a few
                                 microseconds of idle GPU between two short kernels)     */
    /* Column sharding (one process per GPU).  This rank prices the structural columns
     * [col_begin, col_end); 0,0 = all.  See dzg_shard_* below. */
    int64_t col_begin, col_end;
    int32_t rank, world;
    void *stream;             /* hipStream_t to enqueue on; NULL = a private stream.  A host that
                                 interleaves its own collectives passes the stream they are
                                 ordered against (torch.cuda.current_stream().cuda_stream)     */
    int64_t refactor_interval; /* FAST: rebuild the basis inverse from scratch (blocked LU with
                                 partial pivoting, fp64-MFMA trailing updates) every this many
                                 pivots; 0 = never (the eta file is still folded in every 64)  */
    int32_t a_is_block;       /* column sharding, dense: 1 = lp->a holds ONLY this rank's columns
                                 [col_begin, col_end) (column col_begin first), so a rank never
                                 materialises the other ranks' part of the matrix; 0 = lp->a is
                                 the whole m x n_struct matrix and the block is taken from it   */
    int32_t near_tie_action;  /* FAST: dzg_near_tie_action, default COUNT                       */
    int32_t replicate_matrix; /* column sharding, dense: 1 = every rank keeps ALL structural columns in
                                 its HBM and only the PRICING is split by [col_begin, col_end): the
                                 entering column is read locally, so the exchange records shrink to
                                 their 64-byte headers, and a sharded solver can refactorise.  For
                                 matrices that fit one GPU (config 5: 17 GB of 288).  With a_is_block
                                 the other ranks' columns follow through dzg_solver_upload_columns.
                                 0 = partitioned: a rank holds its block only, columns travel in
                                 the records                                                     */
    int32_t auto_restart_rows; /* AUTO: largest LP (rows) that is re-solved in STRICT after FAST met a
                                 near tie / lost its footing; default DZG_AUTO_STRICT_RESTART_ROWS
                                 (2048: 57 ms per STRICT pivot); < 0: never re-solve            */
    double tie_tol;           /* FAST: a decision of the pivot rule is a "near tie" when winner and
                                 runner-up differ by less than max(tie_tol, 64 * max_pivot_error)
                                 relative, or rest on a denominator that is zero up to that
                                 tolerance.  Default 1e-11; < 0 switches the detector off       */
    int32_t seven_launches;   /* FAST, dense matrix resident on the device (one GPU, or a column-sharded
                                 rank with replicate_matrix): 0 (default) = an iteration is three
                                 launches (k_chain_pre, pricing, k_chain_post: device-wide barriers
                                 inside, csrc/k_chain.hip; sharded: the same two kernels between the
                                 exchanges); 1 = the seven launches a partitioned rank runs.  Same
                                 arithmetic, same pivots. */
    int32_t auto_strict_budget_s; /* AUTO: wall-clock seconds the STRICT re-solve may take (31-57 ms per
                                 pivot at 1024-2048 rows: an LP with tens of thousands of pivots
                                 would take an hour).  When it runs out the LP is solved in FAST
                                 numerics with near ties counted, and the result says so
                                 (near_ties > 0, numerics_used = FAST).  0 = default (600 s),
                                 < 0 = no limit */
    int32_t shard_rows;       /* column sharding, dense: 1 = the BASIS SIDE is sharded too -- rank r owns
                                 the rows [r S, (r+1) S), S = ceil4(ceil(m / world)), of x, xbar, dx, of
                                 the compact inverse Binv0 and of the eta columns U; FTRAN, the x-side
                                 ratio test, the eta flush and the update touch a rank's own rows only
                                 (8 m k / world bytes per FTRAN instead of 8 m k on every rank).  Row p of
                                 the inverse (BTRAN) lives on one rank: every x-side candidate travels
                                 with its row in the exchange records (layout below), still two exchanges
                                 per iteration.  x / xbar are complete on every rank whenever
                                 dzg_shard_run / dzg_shard_run_lockstep return.  FAST numerics, pricing
                                 AUTO or TREE.  0 (default): the basis side is replicated on every rank */
    int32_t reserved0;
} dzg_opts;

/* What FAST numerics does at a near tie (the pivot rule is a first-wins strict argmax,
 * src/simplex.rs:432-435,456-459: inside rounding distance of a tie, FAST's rounding and the
 * reference's can disagree, and only the reference's own arithmetic -- STRICT, from the first
 * pivot -- can say which way the reference goes). */
typedef enum {
    DZG_NEAR_TIE_COUNT = 0,   /* carry on; count such pivots in dzg_result.near_ties             */
    DZG_NEAR_TIE_STOP = 1     /* stop with DZG_NEAR_TIE before the pivot is executed             */
} dzg_near_tie_action;

typedef struct {
    int32_t kind;       /* dzg_step_kind           */
    int32_t reserved;
    int64_t entering;   /* variable index j        */
    int64_t leaving;    /* variable index i        */
    double mu;
} dzg_pivot;

/* Kernel classes for the time/byte counters. */
enum {
    DZG_K_STATUS = 0, DZG_K_FTRAN, DZG_K_RATIO, DZG_K_BTRAN, DZG_K_PRICE, DZG_K_UPDATE,
    DZG_K_BASIS_UPDATE, DZG_K_LU,
    /* column-sharded loop (dzg_shard_run): the two all-gathers, from the moment the stream
     * reaches them to their completion (so waiting for a slower rank is counted here).  In that
     * loop STATUS = proposing the first-pivot candidate, FTRAN = the basis kernels before
     * pricing, RATIO = proposing the second record, UPDATE = everything after exchange 2. */
    DZG_K_XCHG1, DZG_K_XCHG2,
    DZG_K_COUNT
};
/* Mirrors of dzg_result in other languages write this number out (the Rust
 * `[c_double; DZG_K_COUNT]`, dantzig_amd/_ffi.py): adding a class is an ABI change. */
#ifdef __cplusplus
static_assert(DZG_K_COUNT == 10, "dzg_result.kernel_ms / kernel_launches have 10 entries");
#else
_Static_assert(DZG_K_COUNT == 10, "dzg_result.kernel_ms / kernel_launches have 10 entries");
#endif

typedef struct {
    int32_t status;          /* dzg_status                                              */
    int32_t numerics_used;   /* STRICT or FAST                                          */
    int64_t iterations;      /* executed pivots                                         */
    double objective;        /* constant + sum c[basis[p]] * x[p]  (src/simplex.rs:345-352) */
    /* optional outputs: NULL = not wanted */
    int64_t *basis;          /* m                                                       */
    int64_t *nonbasis;       /* n-m                                                     */
    double *x, *xbar;        /* m                                                       */
    double *z, *zbar;        /* n-m                                                     */
    dzg_pivot *log;          /* log_cap entries, first min(iterations, log_cap) filled  */
    int64_t log_cap;
    /* counters, filled when opts.profile != 0 */
    double kernel_ms[DZG_K_COUNT];
    int64_t kernel_launches[DZG_K_COUNT];
    double price_bytes;      /* algorithmic bytes of the pricing kernel, summed */
    double solve_ms;         /* wall time inside dzg_solver_run, summed                  */
    double max_pivot_error;  /* FAST health monitor: max relative |dx_p + dz_r| over all pivots
                                (the pivot element computed by FTRAN vs by BTRAN + pricing)   */
    /* FAST near-tie arbitration: the pivot log is the reference's, decision by decision, up to
     * (not including) pivot `first_near_tie`; near_ties == 0 means the whole log is.            */
    int64_t near_ties;       /* executed pivots with a decision inside the tie tolerance        */
    int64_t first_near_tie;  /* iteration index of the first one, -1 if none                    */
    double min_margin;       /* smallest relative margin (winner - runner-up) of any decision   */
    double *margins;         /* optional, like `log`: per-pivot smallest margin, log_cap entries */
    /* FAST basis representation */
    int64_t dense_columns;   /* k: structural variables in the basis = dense columns of the inverse */
    int64_t refactors;       /* refactorisations performed so far                               */
    int64_t chain_fallbacks; /* three-launch iteration: device-wide barriers that failed (another
                                kernel held CUs or LDS of the device) and were recovered from by
                                running on without barriers; 0 on an undisturbed device            */
    int32_t price_pass_used; /* FAST, dense matrix: which pricing pass the executed pivots ran, a bit
                                mask: 1 = row-wise over the k + 1 rows v does not zero (k_price_rows),
                                2 = column-wise over every nonbasic column (k_price_tree / the kernel
                                opts.price_kernel names); 3 = both (k crossed the rule's threshold);
                                0 = no pivot executed, or STRICT / CSC input                       */
    int32_t price_rows_copy; /* 1: the row-major copy of the matrix the row-wise pass streams is
                                resident; 0: not kept -- an explicit price_kernel, CSC, STRICT,
                                DZG_PRICE_ROWS=0, or hipMalloc could not hold a second copy of the
                                matrix (the solve then prices column-wise throughout: same pivots,
                                more bytes early in the solve)                                     */
    double state_drift;      /* FAST: largest relative difference between the carried x_B / z_N and
                                their recomputation from the fresh inverse, measured at the last
                                refactorisation (0 if none): what the near-tie tolerance is widened
                                by (tau = max(tie_tol, 64 max_pivot_error, 4 state_drift))         */
} dzg_result;

typedef struct dzg_solver dzg_solver;

int dzg_abi_version(void);
const char *dzg_status_str(int status);
/* Last HIP / argument error message of the calling thread ("" if none). */
const char *dzg_last_error(void);
/* Number of visible HIP devices (0 if none): lets a host fail early and loudly. */
int dzg_device_count(void);
void dzg_opts_default(dzg_opts *opts);

/* ---- Level 1: replaces Simplex::solve (src/simplex.rs:332-343) -------------------- */

/* Validates the LP, uploads it to HBM and allocates the workspace.  Not timed by bench. */
int dzg_solver_create(const dzg_lp *lp, const dzg_opts *opts, dzg_solver **out);
/* Runs until Optimal / error or until `max_new_iters` more pivots were executed
 * (<= 0: no budget).  Returns the status; DZG_ITER_LIMIT means "budget spent, call again". */
int dzg_solver_run(dzg_solver *s, int64_t max_new_iters);
/* Copies state, pivot log and counters back to the host. */
int dzg_solver_result(dzg_solver *s, dzg_result *res);
void dzg_solver_destroy(dzg_solver *s);
/* opts.replicate_matrix with opts.a_is_block: hands over the structural columns
 * [col_begin, col_end) that the rank did not pass at creation (column col_begin first in `a`,
 * leading dimension lda >= m).  Every column must be present before the first run. */
int dzg_solver_upload_columns(dzg_solver *s, int64_t col_begin, int64_t col_end, const double *a,
                              int64_t lda);
/* FAST: rebuild the basis inverse from scratch now (needs opts.refactor_interval != 0 at
 * creation, which reserves the workspace). */
int dzg_solver_refactor(dzg_solver *s);
/* create + run + result + destroy.  With opts == NULL or numerics AUTO and more than
 * auto_strict_rows rows, FAST runs with near_tie_action = STOP.  Up to DZG_AUTO_STRICT_RESTART_ROWS
 * rows, a run that meets a near tie (or ends in DZG_SINGULAR / DZG_PANIC) is abandoned and the LP
 * is solved again from the first pivot with STRICT numerics -- the reference's arithmetic, the only
 * arbiter of a tie -- and that result is returned.  Above that size STRICT is out of reach
 * (seconds per pivot): FAST carries on and the result reports near_ties / first_near_tie, so the
 * caller knows from which pivot on the path is no longer certified to be the reference's. */
#define DZG_AUTO_STRICT_RESTART_ROWS 2048
int dzg_core_solve(const dzg_lp *lp, const dzg_opts *opts, dzg_result *res);

/* The same on the fields of the reference's `Simplex` exactly as Rust holds them
 * (src/simplex.rs:84-112), so that Simplex::solve (:332-343) becomes ONE call with no
 * reshaping on the Rust side:
 *   constraints: CscMatrix  ->  m = nrows, n = ncols, col_ptr[n+1], row_idx[nnz], val = data[nnz]
 *                               over ALL n columns, slack columns included (src/linalg.rs:161-168;
 *                               Vec<usize> is 64-bit on the reference's targets: pass .as_ptr())
 *   objective               ->  c = coefficients[n], constant
 *   b, n, x, z              ->  basis[m], nonbasis[n-m], x[m], z[n-m]   (x_bar = z_bar = 1, :203-204)
 * The library finds the unit slack columns itself (a column whose only stored entry is 1.0; one per
 * row, scanned from the last column down -- Simplex::new puts them last, :188-201) and keeps the
 * other columns dense or CSC on the device, whichever is smaller.  IN/OUT: on return (status >= 0)
 * basis / nonbasis / x / z hold the final state, as Simplex::solve leaves it in `self`; `res` is
 * filled like dzg_core_solve fills it (its optional buffers may be NULL).  Numerics: as
 * dzg_core_solve (opts == NULL: AUTO). */
int dzg_core_solve_full_csc(int64_t m, int64_t n, const int64_t *col_ptr, const int64_t *row_idx,
                            const double *val, const double *c, double constant, int64_t *basis,
                            int64_t *nonbasis, double *x, double *z, const dzg_opts *opts,
                            dzg_result *res);

/* ---- Level 2: replaces dantzig.rust.solve (src/lib.rs:16-27) ---------------------- */

/* A model as the PyO3 layer hands it over: objective AffExpr (maximised) and a list of
 * Inequality  coef.x <= b  over user variables with optional bounds
 * (src/pyobjs.rs:10-152).  Terms keep expression order: it defines column order and
 * therefore tie-breaks (src/simplex.rs:168-176). */
typedef struct {
    int64_t nvars;
    const int32_t *has_lb, *has_ub;  /* nvars */
    const double *lb, *ub;           /* nvars */
    int64_t obj_nterms;
    const int64_t *obj_var;          /* index into the variable table */
    const double *obj_coef;
    double obj_const;
    int64_t ncons;
    const int64_t *con_ptr;          /* ncons + 1 */
    const int64_t *con_var;
    const double *con_coef;
    const double *con_b;             /* ncons */
} dzg_model;

typedef struct {
    int32_t status;
    int32_t numerics_used;
    int64_t iterations;
    double objective;     /* PySolution.objective_value (core sense: maximised)         */
    double *values;       /* nvars: x+ - x- per user variable (src/simplex.rs:354-371)  */
    int64_t m, n;         /* size of the standard form that was solved                  */
    int64_t near_ties;    /* FAST on an LP too large for a STRICT re-solve: pivots decided inside
                             the tie tolerance (0: the path is the reference's), see dzg_result */
    int64_t first_near_tie;
} dzg_model_result;

int dzg_model_solve(const dzg_model *model, const dzg_opts *opts, dzg_model_result *res);

/* Host-only: the standard-form builder alone (Simplex::new, src/simplex.rs:123-224).
 * Two-call protocol: first call with out->a == NULL fills m, n, n_struct and lda;
 * the caller allocates a[lda*n_struct], var_col[n], c[n], basis[m], nonbasis[n-m],
 * x[m], z[n-m], pos_var[nvars], neg_var[nvars] and calls again. */
typedef struct {
    int64_t m, n, n_struct, lda;
    double *a;
    int64_t *var_col;
    double *c;
    double constant;
    int64_t *basis, *nonbasis;
    double *x, *z;
    int64_t *pos_var, *neg_var;  /* variable index of x+ / x- per user variable, -1 if unseen */
} dzg_stdform;

int dzg_build_standard_form(const dzg_model *model, dzg_stdform *out);

/* ---- single reference functions on the GPU, for parity tests ---------------------- */

/* lu_solve (src/linalg.rs:8-10): a is n*n ROW-major and is not modified; x_out[n].
 * lu_out (n*n, may be NULL) and p_out (n-1, may be NULL) receive the packed factors
 * exactly as Matrix::factorize leaves them (src/linalg.rs:88-128). */
int dzg_kernel_lu_solve(int64_t n, const double *a, const double *b, double *x_out,
                        double *lu_out, int64_t *p_out, int32_t device);

/* collect_columns(cols).neg_t_dot(v) (src/linalg.rs:188-207) on a dense column-major
 * matrix: out[k] = -sum_i a[i, cols[k]] * v[i]; cols[k] < 0 selects the unit column of
 * row (-1 - cols[k]).  kernel: dzg_price_kernel. */
int dzg_kernel_neg_t_dot(int64_t m, int64_t n_struct, const double *a, int64_t lda,
                         const int64_t *cols, int64_t ncols, const double *v, double *out,
                         int32_t kernel, int32_t device);

/* The same on a CSC matrix (the reference's own storage, src/linalg.rs:161-168: rows ascending
 * inside a column, no explicit zeros): out[k] = sum over the stored entries of column cols[k],
 * in stored order, of val * -v[row] -- exactly neg_t_dot's loop (src/linalg.rs:199-207). */
int dzg_kernel_neg_t_dot_csc(int64_t m, int64_t n_struct, const int64_t *col_ptr,
                             const int32_t *row_idx, const double *val, const int64_t *cols,
                             int64_t ncols, const double *v, double *out, int32_t device);

/* find_first_pivot (src/simplex.rs:423-437): position or -1. */
int dzg_kernel_first_pivot(int64_t len, const double *y, const double *ybar, int64_t *pos_out,
                           int32_t device);
/* find_second_pivot (src/simplex.rs:439-461): position or -1. */
int dzg_kernel_second_pivot(int64_t len, double mu, const double *y, const double *ybar,
                            const double *dy, int64_t *pos_out, int32_t device);

/* ---- synthetic LPs of SURVEY 8(d) (host code, SplitMix64; used by bench and tests) -- */

/* Generator G2 (sparse): `per_col` nonzeros per column at distinct rows (sorted ascending),
 * values 2u-1 (redrawn if 0), b = A x0 + rb, c = A^T y0 - rc as in G1.
 * col_ptr[n_struct+1], row_idx[n_struct*per_col], val[n_struct*per_col], b[m], c[n_struct]. */
int dzg_gen_sparse_lp(uint64_t seed, int64_t m, int64_t n_struct, int64_t per_col,
                      int64_t *col_ptr, int32_t *row_idx, double *val, double *b, double *c);

/* Generator G1: dense m x n_struct LP, primal- and dual-feasible by construction.
 * a[lda*n_struct] column-major, b[m], c[n_struct]. */
int dzg_gen_dense_lp(uint64_t seed, int64_t m, int64_t n_struct, double *a, int64_t lda,
                     double *b, double *c);

/* The same LP, bit for bit, but only columns [col_begin, col_end) of A are produced
 * (a_block[lda*(col_end-col_begin)]); b[m] and c[n_struct] are complete.  What one rank of a
 * column-sharded solve needs (pair with opts.a_is_block = 1). */
int dzg_gen_dense_lp_block(uint64_t seed, int64_t m, int64_t n_struct, int64_t col_begin,
                           int64_t col_end, double *a_block, int64_t lda, double *b, double *c);

/* ---- column sharding over several GPUs, one process per GPU ------------------------
 * The host (torch.distributed over RCCL, or any collective layer) moves the small
 * exchange records between ranks; the library only produces and consumes them.
 * See DESIGN.md "Multi-GPU". */
typedef struct {
    double ratio;       /* candidate ratio, -inf when this rank has none                 */
    int64_t pos;        /* GLOBAL nonbasic position of the candidate, INT64_MAX if none   */
    double y, ybar, dy; /* z, zbar, dz of the candidate (for the step lengths s, sbar)    */
} dzg_candidate;

/* A sharded solver is created with opts.world > 1, opts.rank and this rank's column block
 * [col_begin, col_end) (FAST numerics; `a` holds ALL structural columns, only the block is
 * uploaded).  One iteration = three enqueue-only phases with one all-gather of one record per
 * rank between them; record = dzg_shard_record_doubles() doubles:
 *   [0] ratio  [1] global nonbasic position (-1 = none)  [2] z  [3] zbar  [4] dz
 *   [5] column code  [6..7] reserved  [8 .. 8+m) the candidate's column of A
 * phase1: proposes this rank's first-pivot candidate on the z side (src/simplex.rs:275);
 * phase2: merges the proposals (largest ratio, lowest position), runs status() and the part
 *         of the step that precedes the second exchange (primal: FTRAN, x ratio test, BTRAN,
 *         pricing of the owned columns; dual: BTRAN, pricing) and proposes the z-side ratio
 *         candidate (dual) or publishes z, zbar, dz of the entering position (primal);
 * phase3: merges, finishes the step (dual: FTRAN), pivots and updates.
 * `send`/`recv` are DEVICE pointers owned by the host (recv holds world records in rank order).
 * The library never calls a collective itself. */
/* With opts.shard_rows the record is
 *   [0..7]   the z-side candidate as above ([7] reserved)
 *   [8] ratio  [9] basis position p (-1 = none)  [10] x_p  [11] xbar_p  [12] dx_p (second exchange of
 *            a primal step)  [13] reserved  [14] runner-up  [15] reserved      -- the x-side candidate
 *            of the rank's own rows: first pivot (exchange 1), ratio test of a primal step (exchange 2)
 *   [16 .. 80)          U_t[p] of the pending etas t < 64
 *   [80 .. 80 + mc)     the z-side candidate's column of A (mc = ceil16(m); partitioned storage only,
 *                       mc = 0 with replicate_matrix)
 *   [80 + mc .. )       row p of the compact inverse Binv0, as many entries as the record holds
 * dzg_shard_record_doubles is the largest record (room for a row of m entries); dzg_shard_run sends
 * what the current compact width needs.  A host that drives the phases itself keeps the records of
 * exchange 1 intact until phase 3 has been enqueued and run (phase 3 reads the leaving row of a dual
 * step and the entering column of a primal step from them): two receive buffers. */
int64_t dzg_shard_record_doubles(const dzg_solver *s);
int dzg_shard_phase1(dzg_solver *s, double *send_dev);
int dzg_shard_phase2(dzg_solver *s, const double *recv_dev, double *send_dev);
int dzg_shard_phase3(dzg_solver *s, const double *recv_dev);
/* The whole sharded loop in native code.  dzg_shard_comm_init joins this rank to an RCCL
 * communicator (librccl is loaded lazily with dlopen; `unique_id` is the 128-byte ncclUniqueId
 * produced by dzg_comm_unique_id on rank 0 and distributed by the host, e.g. over gloo);
 * dzg_shard_run then iterates phase1 -> ncclAllGather -> phase2 -> ncclAllGather -> phase3 on
 * the solver's stream (xGMI between the GPUs of a node) and polls the status word every
 * poll_interval iterations.  Returns the status like dzg_solver_run. */
int dzg_comm_unique_id(void *unique_id_128);
int dzg_shard_comm_init(dzg_solver *s, const void *unique_id_128);
int dzg_shard_run(dzg_solver *s, int64_t max_new_iters);
/* Ranks of the communicator as RCCL counts them (ncclCommCount), or < 0. */
int dzg_shard_comm_size(dzg_solver *s);
/* All ranks inside one process on one device (solvers[r] created with rank r and a common
 * opts.stream = dzg_solver_stream(solvers[0])): the exchange is a device-to-device copy.
 * This is how the sharded device path is exercised on a single-GPU box. */
int dzg_shard_run_lockstep(dzg_solver **solvers, int32_t world, int64_t max_new_iters);
void *dzg_solver_stream(dzg_solver *s);
/* Changes which kernel classes are timed (opts.profile mask) between runs; the solver must have
 * been created with opts.profile != 0. */
int dzg_solver_set_profile(dzg_solver *s, int32_t mask);
/* Synchronises the stream and reads the status word and the pivot count. */
int dzg_solver_poll(dzg_solver *s, int32_t *status, int64_t *iterations);
/* Sets the run budget like dzg_solver_run does, without running (sharded hosts drive the loop). */
int dzg_solver_set_budget(dzg_solver *s, int64_t max_new_iters);

/* Test hook (tests/test_gpu_parity.py, tools/): occupies `workgroups` CUs of `device` for about
 * `seconds` with a kernel that holds 128 KB of LDS per workgroup and watches the clock, on a
 * stream of its own; returns at once.  dzg_debug_hold_wait waits for it.  This is how the
 * three-launch iteration's recovery from a co-tenant on the device is exercised. */
int dzg_debug_hold_cus(int32_t device, int32_t workgroups, double seconds);
int dzg_debug_hold_wait(void);

/* Test hook: checks the live-entry lists the sparse-basis pricing pass walks (csrc/k_sparse.hip,
 * k_price_csc_rl) against their definition; returns the number of columns whose list is wrong
 * (0 = consistent), < 0 on error (no such lists: DZG_E_ARG); *entries = entries listed in all. */
int64_t dzg_debug_live_lists(dzg_solver *s, int64_t *entries);
/* Test hook: the constraint row whose entries k_sp_btran has listed ahead of a pivot that has not
 * been executed yet (a run that stopped between BTRAN and the pivot, DZG_NEAR_TIE in a dual step's
 * ratio test), -1 if none; < -1 on error.  dzg_debug_live_lists counts that row as listed. */
int64_t dzg_debug_rl_listed(dzg_solver *s);

/* Deterministic max-loc merge: largest ratio wins, lowest global position on ties --
 * the sequential first-wins rule of src/simplex.rs:432-435,456-459.  Returns the index
 * of the winning record, or -1 when every record is empty. */
int64_t dzg_merge_candidates(const dzg_candidate *cands, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* DANTZIG_AMD_H */

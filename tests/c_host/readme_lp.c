/* A plain C host of the drop-in boundary: nothing but include/dantzig_amd.h and the shared
 * library.  It solves the reference README's LP (README.md:60-73)
 *
 *      minimise x + y - z   s.t.  x + y + z == 1,   x, y, z >= 0
 *
 * the way the PyO3 layer would hand it over (Level 2: objective negated for the maximising
 * core, `==` lowered to two opposite inequalities, python-source/dantzig/optimize.py:114-117,
 * model.py:350-375), then a small G1 LP through Level 1 (dzg_core_solve on the post-Simplex::new
 * state), then the README LP once more through dzg_core_solve_full_csc on the fields of the
 * reference's `Simplex` exactly as Rust holds them (src/simplex.rs:84-112: ONE CSC over all
 * columns, slacks included, 64-bit indices) -- the call INTEGRATION.md puts into Simplex::solve.
 * Output: one line per solve, parsed by tests/test_abi.py.
 * Exit code 0 = both solved; 3 = the library reported DZG_E_DEVICE (no GPU: loud failure). */
#include <stdio.h>
#include <stdlib.h>

#include "dantzig_amd.h"

static int level2(void)
{
    const int32_t has_lb[3] = {1, 1, 1}, has_ub[3] = {0, 0, 0};
    const double lb[3] = {0.0, 0.0, 0.0}, ub[3] = {0.0, 0.0, 0.0};
    /* maximise -(x + y - z) */
    const int64_t obj_var[3] = {0, 1, 2};
    const double obj_coef[3] = {-1.0, -1.0, 1.0};
    /*  x + y + z <= 1   and   -x - y - z <= -1 */
    const int64_t con_ptr[3] = {0, 3, 6};
    const int64_t con_var[6] = {0, 1, 2, 0, 1, 2};
    const double con_coef[6] = {1.0, 1.0, 1.0, -1.0, -1.0, -1.0};
    const double con_b[2] = {1.0, -1.0};
    dzg_model model = {3, has_lb, has_ub, lb, ub, 3, obj_var, obj_coef, 0.0,
                       2, con_ptr, con_var, con_coef, con_b};
    double values[3] = {-1.0, -1.0, -1.0};
    dzg_model_result res = {0};
    res.values = values;
    dzg_opts opts;
    dzg_opts_default(&opts);
    const int rc = dzg_model_solve(&model, &opts, &res);
    if (rc < 0) {
        fprintf(stderr, "dzg_model_solve: %s (%s)\n", dzg_status_str(rc), dzg_last_error());
        return rc;
    }
    printf("level2 status=%s iterations=%lld objective=%.17g x=%.17g y=%.17g z=%.17g\n",
           dzg_status_str(res.status), (long long)res.iterations, -res.objective, values[0], values[1],
           values[2]);
    return res.status;
}

static int level1(void)
{
    enum { M = 24, NS = 40, N = M + NS };
    static double a[M * NS], b[M], c_struct[NS], c[N], x[M], z[NS];
    static int64_t basis[M], nonbasis[NS];
    if (dzg_gen_dense_lp(7, M, NS, a, M, b, c_struct) != 0) return DZG_E_ARG;
    for (int j = 0; j < NS; ++j) {
        c[j] = c_struct[j];
        nonbasis[j] = j;
        z[j] = -c_struct[j];
    }
    for (int i = 0; i < M; ++i) {
        c[NS + i] = 0.0;
        basis[i] = NS + i;
        x[i] = b[i];
    }
    dzg_lp lp = {0};
    lp.m = M; lp.n = N; lp.n_struct = NS;
    lp.a = a; lp.lda = M; lp.c = c;
    lp.basis = basis; lp.nonbasis = nonbasis; lp.x = x; lp.z = z;
    static int64_t out_basis[M], out_nonbasis[NS];
    static double out_x[M], out_xbar[M], out_z[NS], out_zbar[NS];
    static dzg_pivot log[4096];
    dzg_result res = {0};
    res.basis = out_basis; res.nonbasis = out_nonbasis;
    res.x = out_x; res.xbar = out_xbar; res.z = out_z; res.zbar = out_zbar;
    res.log = log; res.log_cap = 4096;
    dzg_opts opts;
    dzg_opts_default(&opts);
    opts.numerics = DZG_NUMERICS_STRICT;
    const int rc = dzg_core_solve(&lp, &opts, &res);
    if (rc < 0) {
        fprintf(stderr, "dzg_core_solve: %s (%s)\n", dzg_status_str(rc), dzg_last_error());
        return rc;
    }
    printf("level1 status=%s iterations=%lld objective=%.17g first_pivot=%d:%lld:%lld\n",
           dzg_status_str(res.status), (long long)res.iterations, res.objective,
           res.iterations > 0 ? log[0].kind : -1,
           res.iterations > 0 ? (long long)log[0].entering : -1LL,
           res.iterations > 0 ? (long long)log[0].leaving : -1LL);
    return res.status;
}

/* The README LP after Simplex::new, written the way the reference stores it: the host-only
 * builder gives the dense structural block + column codes; one CSC over all n columns (rows
 * ascending, no explicit zeros, unit slack columns stored like any other) is assembled from it. */
static int level1_full_csc(void)
{
    const int32_t has_lb[3] = {1, 1, 1}, has_ub[3] = {0, 0, 0};
    const double lb[3] = {0.0, 0.0, 0.0}, ub[3] = {0.0, 0.0, 0.0};
    const int64_t obj_var[3] = {0, 1, 2};
    const double obj_coef[3] = {-1.0, -1.0, 1.0};
    const int64_t con_ptr[3] = {0, 3, 6};
    const int64_t con_var[6] = {0, 1, 2, 0, 1, 2};
    const double con_coef[6] = {1.0, 1.0, 1.0, -1.0, -1.0, -1.0};
    const double con_b[2] = {1.0, -1.0};
    dzg_model model = {3, has_lb, has_ub, lb, ub, 3, obj_var, obj_coef, 0.0,
                       2, con_ptr, con_var, con_coef, con_b};
    dzg_stdform sf = {0};
    if (dzg_build_standard_form(&model, &sf) != 0) return DZG_E_ARG; /* sizes */
    enum { CAP = 64 };
    static double a[CAP * CAP], c[CAP], x[CAP], z[CAP], val[CAP * CAP];
    static int64_t var_col[CAP], basis[CAP], nonbasis[CAP], pos_var[3], neg_var[3];
    static int64_t col_ptr[CAP + 1], row_idx[CAP * CAP];
    if (sf.n > CAP || sf.lda > CAP) return DZG_E_ARG;
    sf.a = a; sf.var_col = var_col; sf.c = c; sf.basis = basis; sf.nonbasis = nonbasis;
    sf.x = x; sf.z = z; sf.pos_var = pos_var; sf.neg_var = neg_var;
    if (dzg_build_standard_form(&model, &sf) != 0) return DZG_E_ARG;
    int64_t nnz = 0;
    col_ptr[0] = 0;
    for (int64_t v = 0; v < sf.n; ++v) {
        if (var_col[v] >= 0) {
            for (int64_t i = 0; i < sf.m; ++i) {
                const double e = a[var_col[v] * sf.lda + i];
                if (e != 0.0) { row_idx[nnz] = i; val[nnz] = e; ++nnz; }
            }
        } else {
            row_idx[nnz] = -1 - var_col[v];
            val[nnz] = 1.0;
            ++nnz;
        }
        col_ptr[v + 1] = nnz;
    }
    dzg_result res = {0};
    dzg_opts opts;
    dzg_opts_default(&opts);
    const int rc = dzg_core_solve_full_csc(sf.m, sf.n, col_ptr, row_idx, val, c, sf.constant, basis,
                                           nonbasis, x, z, &opts, &res);
    if (rc < 0) {
        fprintf(stderr, "dzg_core_solve_full_csc: %s (%s)\n", dzg_status_str(rc), dzg_last_error());
        return rc;
    }
    /* Simplex::solution (src/simplex.rs:354-371) on the state that came back in place */
    double user[3] = {0.0, 0.0, 0.0};
    for (int u = 0; u < 3; ++u)
        for (int64_t p = 0; p < sf.m; ++p) {
            if (basis[p] == pos_var[u]) user[u] += x[p];
            if (basis[p] == neg_var[u]) user[u] -= x[p];
        }
    printf("fullcsc status=%s iterations=%lld objective=%.17g x=%.17g y=%.17g z=%.17g m=%lld n=%lld nnz=%lld\n",
           dzg_status_str(res.status), (long long)res.iterations, -res.objective, user[0], user[1],
           user[2], (long long)sf.m, (long long)sf.n, (long long)nnz);
    return res.status;
}

int main(void)
{
    printf("abi %d devices %d\n", dzg_abi_version(), dzg_device_count());
    const int r2 = level2();
    if (r2 == DZG_E_DEVICE) return 3;
    if (r2 != DZG_OPTIMAL) return 1;
    const int r1 = level1();
    if (r1 == DZG_E_DEVICE) return 3;
    if (r1 != DZG_OPTIMAL) return 1;
    const int r3 = level1_full_csc();
    if (r3 == DZG_E_DEVICE) return 3;
    return r3 == DZG_OPTIMAL ? 0 : 1;
}

// k_price.hip -- launchers of the pricing kernels (kernels and design notes: k_price_kernels.h)
#include "k_price_kernels.h"

static_assert(DZG_RL_WORK_SLOTS == DZG_PRICE_CSC_BLOCKS, "work counters of k_price_csc_rl");

static int resolve(int kernel)
{
    return kernel == DZG_PRICE_WAVE || kernel == DZG_PRICE_TREE ? kernel : DZG_PRICE_SEQ;
}

#define PRICE_ARGS ctl, A, lda, m, q, plist, nonbasis, var_col, v, dz, z, zbar, rz_r, rz_k, rz_h, col0

// the tree kernel's sums do not depend on its shape: pick columns-per-wave x tiles-in-flight
// from the number of columns a wave will get at most (ncols is an upper bound of the count)
static void launch_tree(int ncols, const DzgCtl *ctl, const double *A, long long lda, int m, int q,
                        const int *plist, const int *nonbasis, const int *var_col, const double *v,
                        double *dz, const double *z, const double *zbar, double *rz_r, int *rz_k,
                        double *rz_h, int col0, const int *pcode, hipStream_t st, int rows_T = 0,
                        int need_kind = -1)
{
    const dim3 grid(DZG_PRICE_TREE_BLOCKS), block(256);
    const int per_wave = (ncols + 4 * DZG_PRICE_TREE_BLOCKS - 1) / (4 * DZG_PRICE_TREE_BLOCKS);
    // Columns per wave and pass = what a wave actually gets (a load slot of a pass that has no
    // column re-reads the wave's last one: L2 traffic and issue slots for nothing -- one pass of
    // 16 with 12 columns per wave, 12 324 nonbasic structural columns deep in the benchmark solve,
    // streamed at 5.9 TB/s against 6.6 with every slot used, profiles/r03_deep_regime_*).  More
    // than 16 columns: the fewest passes of equal width.  Tiles in flight follow the width so that
    // a wave keeps 26-40 KB on its way, two adjacent tiles per visit of a column where that measured
    // faster (tools/price_width_bench.hip, profiles/r03_price_width_microbench.txt).  The sums do not
    // depend on any of this (only on m).
    const int passes = (per_wave + 15) / 16;
    const int cw = passes > 0 ? (per_wave + passes - 1) / passes : 1;
#define TREE(CW, DEPTH, TP)                                                                          \
    hipLaunchKernelGGL((k_price_tree<CW, DEPTH, TP>), grid, block, 0, st, PRICE_ARGS, pcode, rows_T, need_kind)
    // (15-16 columns per wave: two passes of 8 with 2-KiB visits, 160.8 us against 162.7 for one pass
    // of 16 at 8192 rows, profiles/r02_price_microbench_adjacent_tiles.txt)
    if (per_wave == 15 || per_wave == 16) { TREE(8, 2, 2); return; }
    switch (cw) {
    case 16: TREE(16, 2, 1); break;
    case 15: TREE(15, 2, 1); break;
    case 14: TREE(14, 2, 1); break;
    case 13: TREE(13, 2, 1); break;
    case 12: TREE(12, 3, 1); break;
    case 11: TREE(11, 3, 1); break;
    case 10: TREE(10, 2, 2); break;
    case 9: TREE(9, 2, 2); break;
    case 8: TREE(8, 2, 2); break;
    case 7: TREE(7, 4, 1); break;
    case 6: TREE(6, 3, 2); break;
    case 5: TREE(5, 4, 2); break;
    case 4: TREE(4, 8, 1); break;
    case 3: TREE(3, 10, 1); break;
    case 2: TREE(2, 16, 1); break;
    default: TREE(1, 32, 1); break;
    }
#undef TREE
}

static void launch(int kernel, int ncols, const DzgCtl *ctl, const double *A, long long lda, int m, int q,
                   const int *plist, const int *nonbasis, const int *var_col, const double *v,
                   double *dz, const double *z, const double *zbar, double *rz_r, int *rz_k,
                   double *rz_h, int col0, const int *pcode, hipStream_t st, int rows_T = 0,
                   int need_kind = -1)
{
    if (q <= 0) return;
    if (resolve(kernel) == DZG_PRICE_TREE)
        launch_tree(ncols, PRICE_ARGS, pcode, st, rows_T, need_kind);
    else if (resolve(kernel) == DZG_PRICE_WAVE)
        hipLaunchKernelGGL((k_price_wave2<4>), dim3(DZG_PRICE_WAVE_BLOCKS), dim3(256), 0, st, ctl, A,
                           lda, m, q, plist, nonbasis, var_col, v, dz, z, zbar, rz_r, rz_k, rz_h, col0);
    else
        hipLaunchKernelGGL((k_price_seq2<16>), dim3(DZG_PRICE_SEQ_BLOCKS), dim3(256), 0, st, ctl, A,
                           lda, m, q, plist, nonbasis, var_col, v, dz, z, zbar, rz_r, rz_k, rz_h, col0);
}

// number of per-workgroup ratio partials the chosen kernel leaves in rz_r / rz_k
int dzg_price_partials(int kernel)
{
    if (kernel == DZG_PRICE_CSC_KERNEL) return DZG_PRICE_CSC_BLOCKS; // an upper bound: see below
    if (resolve(kernel) == DZG_PRICE_TREE) return DZG_PRICE_TREE_BLOCKS;
    return resolve(kernel) == DZG_PRICE_WAVE ? DZG_PRICE_WAVE_BLOCKS : DZG_PRICE_SEQ_BLOCKS;
}

// partial count of the pass dzg_launch_price_fast(d, kernel) will run
// workgroups of the live-entry pricing pass (= its ratio partials): 1 024 -- with 2 048 the launch
// after it reduces twice the candidates for nothing (config 4: 15 060 -> 15 870 it/s at k = 1 000,
// 13 250 -> 13 690 at k = 4 126; 512: 15 610 / 13 580); DZG_RL_GRID for A/B
static int rl_grid(void)
{
    static const int g = [] {
        const char *e = std::getenv("DZG_RL_GRID");
        const int v = e ? std::atoi(e) : 1024;
        return v < 1 ? 1 : (v > DZG_PRICE_CSC_BLOCKS ? DZG_PRICE_CSC_BLOCKS : v);
    }();
    return g;
}

int dzg_price_partials_dev(const DzgDev &d, int kernel)
{
    if (!d.csc) return dzg_price_partials(kernel);
    if (d.spb && d.lcnt && kernel != DZG_PRICE_SEQ) return rl_grid();
    return DZG_PRICE_CSC_BLOCKS;
}

static void launch_csc(const DzgDev &d, const int *plist, const double *z, const double *zbar,
                       double *rz_r, int *rz_k, double *rz_h, bool seq_order, hipStream_t st)
{
    if (d.q <= 0) return;
    // FAST (ratio test fused: z != nullptr) sums in tree order unless the caller insists on the
    // reference's order; STRICT always takes the reference's order
    if (z && !seq_order)
        // (column codes per position / per list entry, kept by the pivot's books: one trip each)
        hipLaunchKernelGGL(k_price_csc_tree, dim3(DZG_PRICE_CSC_BLOCKS), dim3(256), 0, st, d.ctl,
                           d.cptr, d.ridx, d.cval, d.q, plist, d.nbcode, (const int *)nullptr, d.v,
                           d.dz, z, zbar, rz_r, rz_k, rz_h, d.col0, plist ? d.pcode : (const int *)nullptr);
    else
        hipLaunchKernelGGL(k_price_csc, dim3(DZG_PRICE_CSC_BLOCKS), dim3(256), 0, st, d.ctl, d.cptr,
                           d.ridx, d.cval, d.q, plist, d.nonbasis, d.var_col, d.v, d.dz, z, zbar,
                           rz_r, rz_k, rz_h, d.col0);
}

// STRICT numerics: every nonbasic position, no fused ratio test
void dzg_launch_price(const DzgDev &d, int kernel, hipStream_t st)
{
    if (d.csc) {
        launch_csc(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, st);
        return;
    }
    launch(kernel, d.q, d.ctl, d.A, d.lda, d.m, d.q, nullptr, d.nonbasis, d.var_col, d.v, d.dz, nullptr,
           nullptr, nullptr, nullptr, nullptr, 0, nullptr, st);
}

// row-major copy of the local structural block for the row-wise pass
void dzg_launch_transpose_to_rows(const double *A, long long lda, int m, int n, double *At,
                                  long long ldt, hipStream_t st)
{
    if (m <= 0 || ldt <= 0) return;
    hipLaunchKernelGGL(k_transpose_to_rows, dim3((unsigned)((ldt + 31) / 32), (unsigned)((m + 31) / 32)),
                       dim3(32, 8), 0, st, A, lda, m, n, At, ldt);
}
int dzg_price_rows_groups(void) { return PR_GMAX; }

// The fused small-k row pass (k_price_rows_small): one GPU, dense matrix with its row-major copy
// and the code -> position map, the host's upper bound on k for the batch below DZG_PRICE_SMALL_K
// (default 480 <= PRS_ROWS - 1; 0: never).
static int price_small_k(void)
{
    static const int v = [] {
        const char *e = std::getenv("DZG_PRICE_SMALL_K");
        const int x = e ? std::atoi(e) : 480;
        return x < 0 ? 0 : (x > PRS_ROWS - 1 ? PRS_ROWS - 1 : x);
    }();
    return v;
}
int dzg_price_small(const DzgDev &d, int kernel)
{
    return d.cpos && d.vc && d.world <= 1 && !d.rs && dzg_price_rows_certain(d, kernel) && d.k_hint < price_small_k() &&
           (d.ldt + PRS_TILE - 1) / PRS_TILE <= 4096;
}
int dzg_price_small_partials(const DzgDev &d) { return (int)((d.ldt + PRS_TILE - 1) / PRS_TILE); }

// 1: the batch being enqueued prices row-wise for certain (dense matrix, row-major copy resident, the
// host's upper bound on k below the rule's threshold): the chain's last launch may then finish the
// pass itself (k_chain_post, FOLD) and the finishing launch is left out
int dzg_price_rows_certain(const DzgDev &d, int kernel)
{
    // (fold_k: an A/B switch, DZG_CHAIN_FOLD_K; the default is "whenever the pass is row-wise")
    return !d.csc && d.At && d.q > 0 && resolve(kernel) == DZG_PRICE_TREE && d.k_hint > 0 &&
           d.k_hint < d.rows_T && d.k_hint < d.fold_k;
}

// FAST numerics: structural positions from plist, ratio-test partials for the dual step
// need_kind >= 0 (row-sharded ranks, dense, tree / row-wise kernels only): the pass runs only in an
// iteration of that step kind
void dzg_launch_price_fast(const DzgDev &d, int kernel, hipStream_t st, int need_kind, int skip_finish,
                           int small)
{
    if (small) { // (the caller asked dzg_price_small)
        // (k < 127 for the whole batch: at most eight row groups, one per wave)
        if (d.k_hint < 8 * DZG_PR_BATCH - 1)
            hipLaunchKernelGGL(k_price_rows_small<false>, dim3(dzg_price_small_partials(d)), dim3(512), 0, st,
                               d.ctl, d.rows_T, d.At, d.ldt, d.col1 - d.col0, d.drow, d.bcode, d.vc, d.cpos, d.q,
                               d.nbcode, d.v, d.dz, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h);
        else
            hipLaunchKernelGGL(k_price_rows_small<true>, dim3(dzg_price_small_partials(d)), dim3(512), 0, st,
                               d.ctl, d.rows_T, d.At, d.ldt, d.col1 - d.col0, d.drow, d.bcode, d.vc, d.cpos, d.q,
                               d.nbcode, d.v, d.dz, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h);
        return;
    }
    if (d.csc) {
        if (d.spb && d.lcnt && kernel != DZG_PRICE_SEQ) { // sparse basis: the live entries only
            if (d.q > 0)
                hipLaunchKernelGGL(k_price_csc_rl, dim3(rl_grid()), dim3(256), 0, st, d.ctl,
                                   d.cptr, d.lcnt, d.lent, d.q, d.plist, d.pcode, d.nbcode, d.v,
                                   d.dz, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h, d.rl_work);
            return;
        }
        launch_csc(d, d.plist, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h, kernel == DZG_PRICE_SEQ, st);
        return;
    }
    // (column codes per nonbasic position, kept by the pivot's books: one load instead of
    // nonbasis[] -> var_col[] before a wave knows where its columns are)
    // pcode[i] = code of the column at plist[i]: a wave learns its columns in one trip
    // the pass shape follows the number of nonbasic structural columns as the host last read it
    // (the kernel takes the exact count from the control block: a stale hint costs speed only)
    int ncols = d.col1 - d.col0;
    if (d.price_cols_hint > 0 && d.price_cols_hint < ncols) ncols = d.price_cols_hint;
    int rows_T = 0;
    if (d.At && d.q > 0 && resolve(kernel) == DZG_PRICE_TREE) {
        // row-wise while k < rows_T (k_price_kernels.h).  The kernels apply the rule themselves; the
        // host's bounds on k (valid for the batch it is enqueueing, else unknown) only spare the
        // launches that cannot apply
        rows_T = d.rows_T;
        const bool rows_possible = d.k_lo_hint < 0 || d.k_lo_hint < rows_T;
        const bool cols_possible = d.k_hint <= 0 || d.k_hint >= rows_T;
        // (whichever pass is left out, the other one prices unconditionally: a bound the host got
        // wrong would cost speed, never a pass)
        const int rows_rule = cols_possible ? rows_T : 0x7fffffff;
        if (!rows_possible) rows_T = 0;
        if (rows_possible) {
            // (two columns per lane, 512 per workgroup: 83 us for k = 4 049 rows of 16 384 columns
            // against 92 with four -- tools/price_rows_bench.hip, profiles/r03_price_rows_microbench.txt)
            hipLaunchKernelGGL((k_price_rows<2>), dim3((unsigned)((d.ldt + 511) / 512), PR_GMAX),
                               dim3(256), 0, st, d.ctl, rows_rule, d.At, d.ldt, d.drow, d.bcode, d.vc,
                               d.ppart, need_kind);
            if (!(skip_finish && !cols_possible))
                hipLaunchKernelGGL(k_price_rows_finish, dim3(DZG_PRICE_TREE_BLOCKS), dim3(256), 0, st,
                                   d.ctl, rows_rule, d.ppart, d.ldt, d.q, d.plist, d.pcode, d.nbcode, d.bcode,
                                   d.col0, d.v, d.dz, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h, PR_GMAX, need_kind);
        }
        if (!cols_possible) return;
    }
    launch(kernel, ncols, d.ctl, d.A, d.lda, d.m, d.q, d.plist, d.nbcode, nullptr, d.v, d.dz, d.z,
           d.zbar, d.rz_r, d.rz_k, d.rz_h, d.col0, d.pcode, st, rows_T, need_kind);
}

void dzg_launch_price_raw(int kernel, int m, long long lda, const double *A, const int *cols,
                          int ncols, const double *v, double *out, hipStream_t st)
{
    launch(kernel, ncols, nullptr, A, lda, m, ncols, nullptr, cols, nullptr, v, out, nullptr, nullptr,
           nullptr, nullptr, nullptr, 0, nullptr, st);
}

// CSC twin of dzg_launch_price_raw: cols[k] >= 0 is a stored column, < 0 the unit column of row
// -1 - cols[k]; no control block, no ratio test.
void dzg_launch_price_csc_raw(const long long *cptr, const int *ridx, const double *cval,
                              const int *cols, int ncols, const double *v, double *out,
                              hipStream_t st)
{
    if (ncols <= 0) return;
    hipLaunchKernelGGL(k_price_csc, dim3(DZG_PRICE_CSC_BLOCKS), dim3(256), 0, st,
                       (const DzgCtl *)nullptr, cptr, ridx, cval, ncols, (const int *)nullptr, cols,
                       (const int *)nullptr, v, out, (const double *)nullptr,
                       (const double *)nullptr, (double *)nullptr, (int *)nullptr,
                       (double *)nullptr, 0);
}

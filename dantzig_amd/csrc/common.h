// common.h -- shared declarations of the gfx950 simplex engine (device + host).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dantzig_amd.h"

#define DZG_WAVE 64

// ---------------------------------------------------------------------------------
// Device-resident control block.  Every per-iteration decision (primal or dual step,
// entering / leaving position, step lengths, termination) is taken on the device and
// read by the following kernels from here, so the host can enqueue iterations blindly
// and only polls `status` every few dozen iterations (no per-pivot host round trip).
// ---------------------------------------------------------------------------------
struct DzgCtl {
    int status;          // dzg_status; kernels are no-ops unless DZG_RUNNING
    int kind;            // dzg_step_kind of the iteration in flight
    long long iter;      // executed pivots
    long long iter_stop; // run budget: status := ITER_LIMIT once iter reaches it
    int enter_pos;       // position in nonbasis[] of the entering variable
    int leave_pos;       // position in basis[] of the leaving variable
    int enter_var;       // variable indices of the pivot just executed (set by k_prepare)
    int leave_var;
    double mu;
    double t, s, tbar, sbar; // step lengths (src/simplex.rs:257-260)
    long long nb_struct; // nonbasic structural columns right now ("s" of SURVEY 8(d))
    double price_bytes;  // algorithmic pricing bytes, accumulated per executed pivot
    // strict LU step scratch
    int lu_mu;
    int lu_pivot_zero;
    // fast numerics: compact basis inverse + pending eta file
    int ncompact;        // dense columns of Binv0 in compact storage ("k")
    int neta;            // pending rank-1 updates not yet folded into Binv0
    int enter_code;      // column code of the entering variable (saves two dependent loads)
    int enter_src;       // sharded: rank whose exchange record carries the entering column
    // sharded: z, zbar, dz of the entering position as published by its owner
    double zr, zbar_r, dz_r;
    int use_record;      // 1: fast_pivot_books takes zr/zbar_r/dz_r instead of its local z arrays
    int del_last;        // >= 0: k_fast_update deletes a compact column of Binv0 (an entering
    int del_ce;          //       slack): column del_ce := column del_last, column del_last := 0
    int rl_listed;       // sparse-basis live-entry lists: the row k_sp_btran has appended to its columns'
                         // lists for a pivot that has not executed yet (-1: none).  A run that stops
                         // between BTRAN and the pivot (DZG_NEAR_TIE in the dual ratio test) leaves
                         // it set, and the resumed iteration does not append the row a second time
    long long nb_nnz;     // sparse mode: stored entries of the owned nonbasic structural columns
    double max_pivot_err; // FAST health: max |dx_p + dz_r| / max(|dx_p|, |dz_r|) since the last
                          // refactorisation
    // FAST near-tie arbitration (see DzgCand2): every argmax of the pivot rule carries its
    // runner-up; a decision whose margin is within `tau` of a tie is one the reference's
    // arithmetic may take the other way.
    double tie_tol;       // opts.tie_tol
    double tau;           // effective tolerance = max(tie_tol, 64 * max_pivot_err)
    double margin;        // smallest relative margin among the decisions of the pivot in flight
    double min_margin;    // ... over all executed pivots
    long long near_ties;  // pivots with at least one decision inside tau
    long long first_near_tie; // iteration index of the first of them, -1: none
    long long tie_skip_iter;  // stop mode: the iteration the host has acknowledged (resume)
    int tie_mode;         // 0: count and carry on, 1: stop with DZG_NEAR_TIE before the pivot
    int tie_seen;         // a decision of the pivot in flight was inside tau
    // sparse-basis mode (k_sparse.hip): what k_sp_pivot booked for k_sp_update to move
    int sp_k;             // k before this pivot
    int sp_app;           // >= 0: row and column appended at this index
    int sp_mrow;          // >= 0: row sp_k-1 of X moves into this row
    int sp_mcol;          // >= 0: column sp_k-1 of X moves into this column
    int sp_zcol;          // >= 0: column slot recycled for another row: cleared
    // dense path: the pivot's books are kept inside the dual-step GEMV launch, whose other
    // workgroups may still be reading neta / ncompact: the new values wait here until
    // k_fast_update (the next launch) commits them
    int neta_next, ncompact_next;
    int fp_count;         // first-pivot partials per side the last update left (its grid size)
    // three-launch chain (k_chain.hip).  k_chain_pre leaves here what k_chain_post's workgroups
    // need about the pivot rows / columns: inside that launch they must not read anything the
    // pivot's books (one workgroup of the same launch) or another workgroup's update rewrites
    int leave_code;       // column code of the leaving variable
    int enter_dslot;      // compact column of the entering slack's row (entering slack only)
    int neta_cur, k_cur;  // neta and ncompact of the iteration in flight
    int bar_timeout;      // a device-wide barrier gave up waiting (status is DZG_PANIC then)
    double xp, xbp;       // x, xbar at the leaving position
    // row-sharded basis side (k_rowshard.hip): the leaving row's owner and dx at the leaving position
    // as the records / the replicated row arithmetic delivered them
    int leave_src;        // rank whose record carries row p of the inverse (exchange 1: dual step,
                          // exchange 2: primal step)
    int price_mask;       // pricing passes the executed pivots ran: 1 row-wise, 2 column-wise
    double dxp;
    double drift_tau;     // 4 x the relative drift of the carried x, xbar, z against the fresh inverse
                          // at the last refactorisation (k_drift.hip); part of tau
    unsigned long long bar_gen; // device-wide barriers passed so far (k_chain.hip)
};

// Partial-reduction fan-in sizes of the FAST pipeline (fixed grids => fixed counts)
#define DZG_NB_UPD 64     // blocks of k_fast_update      -> first-pivot partials per side
#define DZG_NB_GEMV 512   // blocks of k_fast_gemv        -> primal ratio partials
#define DZG_NW_PRICE 1024 // waves of the pricing kernels -> dual ratio partials
#define DZG_RMAX 64       // eta-file capacity = rank of one MFMA flush
#define DZG_PR_GMAX 32    // row groups of the row-wise pricing pass at most (k_price_rows)
#define DZG_PR_BATCH 16   // rows per group at least: G = min(DZG_PR_GMAX, ceil((k + 1) / 16))
// exchange record of a row-sharded rank (include/dantzig_amd.h): header, U_t[p], column, row
#define DZG_RS_HDR 16
#define DZG_RS_COL (DZG_RS_HDR + DZG_RMAX)

// argmax candidate: k < 0 means "none"
struct DzgCand {
    double r;
    int k;
};

// FAST numerics: argmax candidate that also carries its competition.  `h` is the largest
// ratio among all OTHER candidates that took part (the runner-up), -inf when there was none,
// +inf when some candidate's ratio cannot be trusted at all (a denominator that is zero up to
// rounding: the reference may have +-inf, NaN or an excluded candidate there).
struct DzgCand2 {
    double r;
    int k;
    double h;
};

#ifdef __HIPCC__

__device__ __forceinline__ DzgCand2 dzg_cand2_none()
{
    DzgCand2 c;
    c.r = 0.0;
    c.k = -1;
    c.h = -__builtin_inf();
    return c;
}

// Same winner as dzg_better; the loser's ratio joins the competition record.
__device__ __forceinline__ DzgCand2 dzg_better2(DzgCand2 a, DzgCand2 b)
{
    const bool b_wins = a.k < 0 ? true : (b.k < 0 ? false : (b.r > a.r || (b.r == a.r && b.k < a.k)));
    DzgCand2 w = b_wins ? b : a;
    const DzgCand2 l = b_wins ? a : b;
    double h = a.h > b.h ? a.h : b.h;
    if (l.k >= 0 && l.r > h) h = l.r;
    w.h = h;
    return w;
}

__device__ __forceinline__ DzgCand2 dzg_wave_best2(DzgCand2 c)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        DzgCand2 o;
        o.r = __shfl_xor(c.r, off, DZG_WAVE);
        o.k = __shfl_xor(c.k, off, DZG_WAVE);
        o.h = __shfl_xor(c.h, off, DZG_WAVE);
        c = dzg_better2(c, o);
    }
    return c;
}

// Block-wide reduction; result valid in every thread.  blockDim.x <= 1024.
__device__ __forceinline__ DzgCand2 dzg_block_best2(DzgCand2 c)
{
    __shared__ double s2_r[16], s2_h[16];
    __shared__ int s2_k[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    c = dzg_wave_best2(c);
    __syncthreads(); // protect the arrays from a previous use
    if (lane == 0) {
        s2_r[wave] = c.r;
        s2_k[wave] = c.k;
        s2_h[wave] = c.h;
    }
    __syncthreads();
    DzgCand2 o = dzg_cand2_none();
    if (lane < nw) {
        o.r = s2_r[lane];
        o.k = s2_k[lane];
        o.h = s2_h[lane];
    }
    return dzg_wave_best2(o);
}

// Relative margin of a decision: (winner - runner-up) / max(|winner|, |runner-up|).
// +inf: no competition; 0: exact tie; -1: the decision rests on an untrustworthy ratio.
__device__ __forceinline__ double dzg_margin(DzgCand2 c)
{
    const double inf = __builtin_inf();
    if (c.h == inf) return -1.0;
    if (c.k < 0 || c.h == -inf) return inf;
    const double a = fabs(c.r), b = fabs(c.h);
    const double den = a > b ? a : b;
    if (!(den > 0.0)) return 0.0;
    if (den == inf) return c.r == c.h ? 0.0 : inf; // +inf ratios are legitimate
    return (c.r - c.h) / den;
}

// A zero that is only zero up to rounding: |d| <= tau * (|a| + |b| + 1) for d = a + b.
__device__ __forceinline__ bool dzg_noise_zero(double d, double a, double b, double tau)
{
    return fabs(d) <= tau * (fabs(a) + fabs(b) + 1.0);
}

__device__ __forceinline__ DzgCand dzg_better(DzgCand a, DzgCand b)
{
    // Largest ratio wins; on equal ratios the LOWER position wins.  This is the parallel
    // form of the reference's sequential "replace only if ratio > best" scan
    // (src/simplex.rs:432-435, :456-459); 0.0 and -0.0 compare equal, as there.
    if (b.k < 0) return a;
    if (a.k < 0) return b;
    if (b.r > a.r || (b.r == a.r && b.k < a.k)) return b;
    return a;
}

__device__ __forceinline__ DzgCand dzg_wave_best(DzgCand c)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        DzgCand o;
        o.r = __shfl_xor(c.r, off, DZG_WAVE);
        o.k = __shfl_xor(c.k, off, DZG_WAVE);
        c = dzg_better(c, o);
    }
    return c;
}

// Block-wide reduction; result valid in every thread.  blockDim.x <= 1024.
__device__ __forceinline__ DzgCand dzg_block_best(DzgCand c)
{
    __shared__ double s_r[16];
    __shared__ int s_k[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    c = dzg_wave_best(c);
    __syncthreads(); // protect s_r/s_k from a previous use
    if (lane == 0) {
        s_r[wave] = c.r;
        s_k[wave] = c.k;
    }
    __syncthreads();
    DzgCand o;
    o.r = (lane < nw) ? s_r[lane] : 0.0;
    o.k = (lane < nw) ? s_k[lane] : -1;
    o = dzg_wave_best(o);
    return o;
}

// IEEE-754 division, correctly rounded.  The compiler's fp64 division sequence for gfx950
// (v_div_scale / v_rcp / Newton steps / v_div_fmas / v_div_fixup) is off by one unit in the last
// place when the exact quotient lies within ~1e-32 (relative) of a rounding boundary -- e.g.
// 1.3999999999999992 / -0.9999999999999994 -- which never shows on random data and does show on
// "decimal" data (fuzz seed 2259, tools/fuzz_parity.py).  x86 (the reference's hardware) rounds
// correctly, so every division a decision or a stored value depends on goes through here: the
// residual of a quotient that is within an ulp is exact in one FMA, and so is the neighbour's;
// the smaller residual is the nearer quotient.
__device__ __forceinline__ double dzg_div(double a, double b)
{
    double q = a / b;
    const double aa = fabs(a), ab = fabs(b), aq = fabs(q);
    if (aa > 0x1p-900 && aa < 0x1p900 && ab > 0x1p-900 && ab < 0x1p900 && aq > 0x1p-900 &&
        aq < 0x1p900) { // no overflow / underflow in the residuals
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) { // one step repairs an error of one ulp; two, of two
            const double r = fma(-q, b, a);
            if (r == 0.0) break;
            const bool towards_plus = (r > 0.0) == (b > 0.0); // the exact quotient is above q
            const bool grow = towards_plus == (q > 0.0);      // its magnitude is larger than |q|
            const double q2 = __longlong_as_double(__double_as_longlong(q) + (grow ? 1 : -1));
            const double r2 = fma(-q2, b, a);
            if (!(fabs(r2) < fabs(r))) break;
            q = q2;
        }
    }
    return q;
}

// One entry of find_first_pivot (src/simplex.rs:423-437: ybar_k > 0 makes a candidate, ratio
// -y_k / ybar_k) joins a FAST argmax.  Entries whose ybar is zero to within FAST's rounding (tau):
//   y not above tau either: a 0/0-like ratio -- the side is marked untrustworthy (h = +inf), as ever;
//   y above tau: the ratio is hugely negative and never wins an argmax, but WHETHER the entry is a
//     candidate at all is the sign of a rounding error -- and the reference's status() takes another
//     branch when a side has no candidate (src/simplex.rs:299-303).  Such an entry stands aside as a
//     pseudo-candidate of ratio -inf: every real candidate beats it, and a side whose best is a
//     pseudo-candidate is treated as empty AND flagged by fast_status (found by the fuzz, seed 40037).
__device__ __forceinline__ void dzg_first_pivot_entry(DzgCand2 &best, double y, double yb, int idx, double tau)
{
    const double inf = __builtin_inf();
    const bool noise = fabs(yb) <= tau;
    if (noise && y > tau) {
        DzgCand2 c;
        c.r = -inf;
        c.k = idx;
        c.h = -inf;
        best = dzg_better2(best, c);
        return;
    }
    if (yb > 0.0) {
        DzgCand2 c;
        c.r = dzg_div(-y, yb);
        c.k = idx;
        c.h = -inf;
        if (c.r == c.r) best = dzg_better2(best, c);
    }
    if (noise) best.h = inf;
}


// x / y with 0 / 0 = 0 (src/simplex.rs:464-468); *ok cleared on a non-finite result.
__device__ __forceinline__ double dzg_safe_divide(double x, double y, int *ok)
{
    double d = (x == 0.0 && y == 0.0) ? 0.0 : dzg_div(x, y);
    if (isinf(d) || isnan(d)) *ok = 0;
    return d;
}

__device__ __forceinline__ double dzg_readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

#endif // __HIPCC__

// ---------------------------------------------------------------------------------
// STRICT LU workspace (k_strict.hip): blocked right-looking LU in the reference's operation order
// ---------------------------------------------------------------------------------
#define DZG_LU_NB 64  // panel width
#define DZG_LU_LDP 66 // panel buffer row stride: 64 matrix columns, the right-hand side at [64]
struct DzgLu {
    int n;
    long long ldw;   // row stride of W; columns 0..n-1 = matrix, column n = right-hand side
    double *W;       // n x ldw, row-major
    double *P0, *P1; // n x DZG_LU_LDP each: the active part of the panel, ping-pong
    int *piv;        // n   pivot row of step k (Matrix::factorize's p)
    int *pz;         // n   1 if the pivot of step k was exactly zero (step skipped)
    int *ptab;       // DZG_LU_NB x n   row positions per step (see k_strict.hip)
    double *part_r;  // 2 x nparts   per-workgroup pivot-search maxima, ping-pong
    int *part_k;     // 2 x nparts
    int nparts;
};
void dzg_lu_layout(int n, DzgLu *w); // fills n, ldw, nparts

// ---------------------------------------------------------------------------------
// Host-side launch wrappers (defined in the .hip files, called by engine.hip).
// All take the stream; all kernels early-out unless ctl->status == DZG_RUNNING.
// ---------------------------------------------------------------------------------
// one live entry of a column's list (k_price_csc_rl)
struct __attribute__((aligned(16))) DzgLiveEntry {
    int row;
    int pad_;
    double val;
};

struct DzgDev {
    // problem
    int m, q, n, ns;
    long long lda;
    const double *A;   // column-major m x ns, lda (dense mode)
    // sparse mode (csc != 0): the owned structural columns in CSC, rows ascending per column
    int csc;
    const long long *cptr; // [col1 - col0 + 1]
    const int *ridx;       // [nnz]
    const double *cval;    // [nnz]
    const int *var_col; // n
    // state
    int *basis, *nonbasis; // m, q
    double *x, *xbar, *z, *zbar; // m, m, q, q
    double *dx, *dz, *v, *acol; // m, q, m (+2 zero pads), m
    DzgCtl *ctl;
    // log
    int *log_kind, *log_enter, *log_leave;
    double *log_mu;
    long long log_cap;
    // fast numerics (k_fast.hip): Binv = Binv0 - U W^T
    double *binv;      // Binv0, compact row-major: m rows x ncompact dense columns, stride ldb
    long long ldb;
    int *drow;         // [m] constraint row of compact column c
    int *dslot;        // [m] compact column of constraint row r, or -1 (unit column)
    double *U;         // [DZG_RMAX][ldw] eta columns u_t (each contiguous)
    double *W;         // [DZG_RMAX][ldw] eta rows (each W_t contiguous)
    long long ldw;
    double *Wc;        // [DZG_RMAX][ldw] W gathered to compact coordinates (flush scratch)
    double *ag;        // [m] entering column gathered to compact coordinates
    double *beta;      // [DZG_RMAX] W_t . a_j
    int *plist;        // [q] nonbasic positions holding structural variables (first nb_struct)
    int *pslot;        // [q] index into plist or -1
    int *pcode;        // [q] column code of the variable at plist[i] (the pricing waves' column list)
    int *cpos;         // [col1 - col0] nonbasic position of structural column col0 + j, -1 while it is basic
                       // (dense matrix, one GPU: k_price_rows_small walks column tiles, not positions)
    double *fpx_r, *fpz_r, *rx_r, *rz_r; // partial candidates (ratios)
    int *fpx_k, *fpz_k, *rx_k, *rz_k;    // partial candidates (positions)
    double *fpx_h, *fpz_h, *rx_h, *rz_h; // partial candidates (runner-up ratios, DzgCand2::h)
    double *log_margin;                  // [log_cap] smallest decision margin of each pivot
    int *bcode, *nbcode;                 // [m], [q] column code of the variable at each basis /
                                         // nonbasis position (var_col[basis[p]]: one load, not two)
    // sparse-basis mode (spb != 0: CSC input on one GPU, k_sparse.hip): binv holds X = the k x k
    // block of B^-1 (row b = structural basis position spos[b], column c = constraint row drow[c]),
    // U and W the pending etas in the same compact numbering
    int spb;
    const long long *rptr; // CSR copy of the structural block: [m + 1]
    const int *cidx;       // [nnz] column of each stored entry, ascending inside a row
    const double *rval;    // [nnz]
    int *sslot;            // [m]  row of X of basis position p, -1: a slack is basic there
    int *spos;             // [m]  basis position of row b of X
    int *bslot;            // [ns] row of X of structural column j when it is basic, else -1
    int *rowpos;           // [m]  basis position of the slack of constraint row r, -1: nonbasic
    int *bcnt;             // [m]  entries of constraint row r in BASIC structural columns ...
    int *bcol;             // [nnz] ... their columns, in row r's CSR slice (rptr[r] + i) ...
    double *bval;          // [nnz] ... and values
    double *dxs;           // [m]  dx on the structural positions, by row of X
    int *acol_code;        // column code currently scattered in acol (INT_MIN: none)
    // live entries of every structural column (rows of R: the rows whose slack is nonbasic), in the
    // column's CSC slice; NULL: not kept (price_kernel = SEQ, DZG_SP_PRICE_FULL=1)
    int *lcnt;             // [ns]  live entries of column j ...
    DzgLiveEntry *lent;    // [nnz] ... row and value of each (16 bytes: one load, one cache line
                           //       for a short list), at cptr[j] + i, in the order the rows joined R
    unsigned long long *rl_work; // [DZG_PRICE_CSC_BLOCKS] entries k_price_csc_rl walked, per workgroup
    // strict numerics
    DzgLu lu;
    double eps;
    // column sharding (world > 1): this rank holds structural columns [col0, col1)
    int col0, col1, rank, world;
    int repl;            // 1: every structural column is resident (A points at column col0 of the
                         //    whole matrix, so `A + (code - col0) * lda` reaches any column)
    long long xstride;   // exchange record stride in doubles (8: header only, repl)
    // row-wise pricing of a dense matrix (k_price_rows, k_price_kernels.h); At == NULL: not kept
    const double *At;    // row-major copy of the local structural block: m rows of ldt
    long long ldt;       // >= col1 - col0, a multiple of 4; the padding columns are zero
    double *ppart;       // [PR_GMAX][ldt] partial sums of the row groups
    double *vc;          // [m] v in compact numbering (vc[c] = v[drow[c]]), written by BTRAN
    int rows_T;          // the row-wise pass prices an iteration while ctl->ncompact < rows_T
    int k_lo_hint;       // a LOWER bound of ctl->ncompact for the batch being enqueued; -1: unknown
    int k_hint;          // an UPPER BOUND of ctl->ncompact while the enqueued batch runs (the host's
                         // last reading + the batch: k grows by at most one per pivot); 0: unknown
                         // (= m).  Sizes the eta flush's grid: k is only known on the device, and a
                         // grid for k = m is 611 000 workgroups at 50 000 rows, all but (k / 64)^2
                         // of them empty -- 600 us per flush, 9 us per pivot
    int price_cols_hint; // host's last reading of ctl->nb_struct (0: unknown): picks the pricing
                         // kernel's pass shape; any shape is correct for any count
    // opts.shard_rows (k_rowshard.hip): this rank owns the rows [rs_r0, rs_r1) of x, xbar, dx, Binv0
    // and U; rs == 0: all of them (rs_r0 = 0, rs_r1 = m)
    int rs, rs_r0, rs_r1;
    long long rs_mcol;   // doubles of a record's column section (0: the matrix is replicated)
    int ftran_variant;   // how FTRAN loads the rows of Binv0 (fast_gemv_row_head<.., VAR>): >= 0 forced
                         // (DZG_FTRAN_VARIANT), -1: nontemporal from k = ftran_nt_k on, plain below
    int ftran_nt_k;      // 8 m k bytes of inverse beyond 1.5 x the 256-MB Infinity Cache
    int fold_k;          // k_chain_post finishes the row-wise pricing pass itself while k stays below
                         // this (default: always; DZG_CHAIN_FOLD_K is the A/B switch)
};

#ifdef __HIPCC__
// column of the entering variable: the local matrix (single GPU) or the winner's exchange
// record (sharded).  code < 0 (slack) has no stored column.
__device__ __forceinline__ const double *dzg_enter_col(const DzgCtl *ctl, int code,
                                                       const double *A, long long lda, int col0,
                                                       const double *xrecv, long long xstride)
{
    if (code < 0) return nullptr;
    // header-only records (xstride == 8): the matrix is replicated, every column is local
    if (xrecv && xstride > 8) return xrecv + (long long)ctl->enter_src * xstride + 8;
    return A + (long long)(code - col0) * lda;
}
#endif

// k_vector.hip
void dzg_launch_status(const DzgDev &d, hipStream_t st);
void dzg_launch_ratio(const DzgDev &d, int need_kind, hipStream_t st);
void dzg_launch_prepare(const DzgDev &d, hipStream_t st);
void dzg_launch_update_vectors(const DzgDev &d, hipStream_t st);
void dzg_launch_load_column(const DzgDev &d, int need_kind, hipStream_t st);
void dzg_launch_unit_rhs(const DzgDev &d, hipStream_t st);
int dzg_run_first_pivot(int64_t len, const double *y, const double *ybar, int64_t *pos_out);
int dzg_run_second_pivot(int64_t len, double mu, const double *y, const double *ybar,
                         const double *dy, int64_t *pos_out);

// k_price.hip
void dzg_launch_price(const DzgDev &d, int kernel, hipStream_t st);
void dzg_launch_price_fast(const DzgDev &d, int kernel, hipStream_t st, int need_kind = -1,
                           int skip_finish = 0, int small = 0);
int dzg_price_small(const DzgDev &d, int kernel);       // 1: the batch runs the fused small-k row pass
int dzg_price_small_partials(const DzgDev &d);          // its workgroups (= ratio partials)
int dzg_price_rows_certain(const DzgDev &d, int kernel); // the host's bounds on k prove the row-wise pass
void dzg_launch_transpose_to_rows(const double *A, long long lda, int m, int n, double *At,
                                  long long ldt, hipStream_t st);
int dzg_price_rows_groups(void);
int dzg_price_partials(int kernel);
int dzg_price_partials_dev(const DzgDev &d, int kernel);
#define DZG_PRICE_CSC_KERNEL 100 // internal id: the CSC pricing kernel
#define DZG_RL_WORK_SLOTS 2048   // workgroups of k_price_csc_rl (= its per-workgroup work counters)
void dzg_launch_price_raw(int kernel, int m, long long lda, const double *A, const int *cols,
                          int ncols, const double *v, double *out, hipStream_t st);
void dzg_launch_price_csc_raw(const long long *cptr, const int *ridx, const double *cval,
                              const int *cols, int ncols, const double *v, double *out,
                              hipStream_t st);

// k_strict.hip
// solves  B y = rhs  (transposed == 0, rhs = d.acol -> d.dx)  or
//         B^T y = rhs (transposed == 1, rhs = unit(leave_pos) -> d.v)
// with the reference's dense LU; `need_kind` < 0 runs unconditionally.
void dzg_launch_strict_solve(const DzgDev &d, int transposed, hipStream_t st);
void dzg_launch_lu_raw(const DzgLu &w, DzgCtl *ctl, double *x_out, hipStream_t st);

// k_fast.hip
void dzg_launch_fast_init(const DzgDev &d, hipStream_t st);
void dzg_launch_fast_select_prep(const DzgDev &d, int mode, int nrz, const double *xrecv,
                                 hipStream_t st);
void dzg_launch_fast_gemv(const DzgDev &d, int need_kind, const double *xrecv, hipStream_t st);
void dzg_launch_fast_btran(const DzgDev &d, hipStream_t st);
void dzg_launch_fast_update(const DzgDev &d, int only_partials, hipStream_t st);
void dzg_launch_fast_flush(const DzgDev &d, hipStream_t st);
void dzg_launch_refactor_lists(const DzgDev &d, int *spos, int *scode, int *lpos, int *lrow,
                               int *counts, hipStream_t st);
// the refactorisation in three stages (k_refactor.hip): between A and B, and between B and C, a
// partitioned column-sharded solve sums G (k x ldg) and the block B returns (nl x ldg) over its ranks
void dzg_launch_refactor_a(const DzgDev &d, int k, double *G, double *X, long long ldg, int *scode,
                           int *singular, hipStream_t st);
double *dzg_launch_refactor_b(const DzgDev &d, int k, int nl, double *G, double *X, double *Pn, double *Tri,
                              long long ldg,
                              int *piv, int *spos, int *scode, int *lpos, int *lrow, int *lslot,
                              int *singular, hipStream_t st);
void dzg_launch_refactor_c(const DzgDev &d, int k, int nl, double *G, double *X, long long ldg,
                           int *lpos, int *singular, hipStream_t st);
void dzg_launch_lockstep_sum(double *const *bufs, int world, long long count, hipStream_t st);
void dzg_launch_shard_propose(const DzgDev &d, int mode, int nrz, double *xsend, hipStream_t st);
void dzg_launch_lockstep_allgather(double *const *ptrs, int world, int which, long long xstride,
                                   hipStream_t st);

// k_chain.hip: the FAST iteration of the dense inverse in three launches (one GPU)
#define DZG_CHAIN_AGCAP 16384 // compact width up to which the chain runs (the gathered column in LDS)
#define DZG_CHAIN_BAR_WORDS (16 * 10) // barrier counters (chain_barrier.h: CH_BAR_WORDS)
int dzg_chain_resident_per_cu(void); // workgroups of the chain kernels the runtime places on one CU
void dzg_launch_chain_pre(const DzgDev &d, int grid, unsigned long long *bar,
                          unsigned long long *dbg, const double *xrecv, hipStream_t st);
void dzg_launch_chain_post(const DzgDev &d, int grid, unsigned long long *bar,
                           unsigned long long *dbg, int only_partials, int nrz, const double *xrecv,
                           hipStream_t st, int fold = 0, int price_small = 0);

// k_rowshard.hip: column sharding with the basis side sharded by rows too (opts.shard_rows)
void dzg_launch_rs_propose(const DzgDev &d, int mode, int nrz, double *xsend, hipStream_t st);
void dzg_launch_rs_select(const DzgDev &d, int mode, const double *xrecv, hipStream_t st);
void dzg_launch_rs_gemv(const DzgDev &d, int kind, const double *xrecv1, const double *xrecv2,
                        hipStream_t st);
void dzg_launch_rs_books(const DzgDev &d, const double *xrecv1, hipStream_t st);
void dzg_launch_rs_lockstep_gather(double *const *ptrs, int world, int m, int slice, hipStream_t st);
void dzg_launch_rs_pack(const DzgDev &d, int slice, double *send, hipStream_t st);
void dzg_launch_rs_unpack(const DzgDev &d, int slice, const double *recv, hipStream_t st);

// k_drift.hip: carried state against the fresh inverse, at a refactorisation
void dzg_launch_drift(const DzgDev &d, const double *b0, const double *xb0, const double *cdev,
                      double *agb, double *agx, double *part, double *y, double *dzy, double *out,
                      int k_bound, hipStream_t st);
int dzg_drift_blocks(void);
int dzg_drift_chunks(void);

// k_sparse.hip
void dzg_launch_sp_init(const DzgDev &d, int first, hipStream_t st);
void dzg_launch_sp_ftran(const DzgDev &d, int need_kind, int nrz, hipStream_t st);
void dzg_launch_sp_btran(const DzgDev &d, hipStream_t st);
void dzg_launch_sp_pivot(const DzgDev &d, hipStream_t st);
void dzg_launch_sp_update(const DzgDev &d, int only_partials, hipStream_t st);
void dzg_launch_sp_flush(const DzgDev &d, hipStream_t st);
void dzg_launch_sp_ref_copy(const DzgDev &d, int k, const double *Xinv, long long ldx, hipStream_t st);
// the same iteration in four launches (k_sp_pre, pricing, k_sp_mid, k_sp_update)
int dzg_sp_grid(int m);
int dzg_sp_fused_resident_per_cu(void);
void dzg_launch_sp_pre(const DzgDev &d, unsigned long long *bar, hipStream_t st);
void dzg_launch_sp_mid(const DzgDev &d, unsigned long long *bar, int nrz, hipStream_t st);

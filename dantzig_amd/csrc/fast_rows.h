// fast_rows.h -- device code shared by the FAST kernels of the dense inverse: the seven-launch
// iteration (k_fast.hip; column-sharded solvers run it between their exchanges) and the
// three-launch chain (k_chain.hip; one GPU).  Both call the SAME row, dot-product and book-keeping
// functions, so a solve is bit-identical whichever of the two runs an iteration.
#pragma once
#include "common.h"
#include "fast_decide.h"

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

#define R_ DZG_RMAX

__device__ __forceinline__ double block_sum(double x)
{
    __shared__ double s_sum[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, DZG_WAVE);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_sum[w];
    return t;
}

// Merge of the ranks' proposals.  Slack positions are replicated, so several ranks may propose
// the SAME position (with the same ratio): that is one candidate, not a tie.
__device__ __forceinline__ int shard_merge(const double *__restrict__ xrecv, long long xstride,
                                           int world, DzgCand2 &win)
{
    int w = -1;
    win = dzg_cand2_none();
    for (int r = 0; r < world; ++r) {
        const double *rec = xrecv + (long long)r * xstride;
        DzgCand2 c;
        c.r = rec[0];
        c.k = (int)rec[1];
        c.h = rec[6]; // the rank's own runner-up (or hazard mark)
        if (c.k < 0 || c.r != c.r || c.k == win.k) {
            if (c.h > win.h) win.h = c.h;
            continue;
        }
        win = dzg_better2(win, c);
        if (win.k == c.k) w = r;
    }
    return w;
}

// beta_t = W_t . a_j by one workgroup (every thread calls; the first 256 threads carry the sum, in
// an order that depends on m only: four strided partial sums per thread, a wave tree, the waves in
// order -- further waves add +0.0, so workgroups of 256 and of 512 threads give the same bits)
__device__ __forceinline__ double fast_beta_dot(const double *__restrict__ wt,
                                                const double *__restrict__ a, int m)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (threadIdx.x < 256) {
        int i = threadIdx.x;
        // (unrolled: the loads of eight steps leave together -- after a kernel boundary each is a
        // trip to HBM -- and the sums are then taken in the same order as ever)
#pragma unroll 8
        for (; i + 3 * 256 < m; i += 4 * 256) {
            a0 = fma(wt[i], a[i], a0);
            a1 = fma(wt[i + 256], a[i + 256], a1);
            a2 = fma(wt[i + 512], a[i + 512], a2);
            a3 = fma(wt[i + 768], a[i + 768], a3);
        }
        for (; i < m; i += 256) a0 = fma(wt[i], a[i], a0);
    }
    return block_sum((a0 + a1) + (a2 + a3));
}

// One row of dx = Binv a_j by the LPR lanes of a wave that share `i` (lane `sub` of LPR), in two
// steps so that a kernel may do the first before beta is known: (head) this lane's share of the
// compact row of Binv0 against the gathered column `ag` (padded with a zero to an even length);
// (tail) minus this lane's share of the eta file, then the sum over the LPR lanes.  The summation
// order depends on LPR, k and neta only.  Rows i >= m: zero (the lanes still take part in the tail).
// VAR (the loads of the row only; the sums and their order are the same in every variant):
//   0  plain loads, four steps' loads in flight          1  nontemporal loads
//   2  nontemporal loads, eight steps' loads in flight
template <int LPR, int VAR = 0>
__device__ __forceinline__ double fast_gemv_row_head(int i, int m, int k2,
                                                     const double *__restrict__ binv, long long ldb,
                                                     const double *__restrict__ ag, int sub)
{
    if (i >= m) return 0.0;
    const double *row = binv + (long long)i * ldb;
    double a0 = 0.0, a1 = 0.0;
    int c = 2 * sub;
    if (VAR != 0) {
        constexpr int UN = VAR == 2 ? 8 : 4;
#pragma unroll UN
        for (; c + 2 * LPR < k2; c += 4 * LPR) {
            const double2_t r0 = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(row + c));
            const double2_t r1 = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(row + c + 2 * LPR));
            const double2_t g0 = *reinterpret_cast<const double2_t *>(ag + c);
            const double2_t g1 = *reinterpret_cast<const double2_t *>(ag + c + 2 * LPR);
            a0 = fma(r0.x, g0.x, a0);
            a1 = fma(r1.x, g1.x, a1);
            a0 = fma(r0.y, g0.y, a0);
            a1 = fma(r1.y, g1.y, a1);
        }
        for (; c < k2; c += 2 * LPR) {
            const double2_t r0 = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(row + c));
            const double2_t g0 = *reinterpret_cast<const double2_t *>(ag + c);
            a0 = fma(r0.x, g0.x, a0);
            a0 = fma(r0.y, g0.y, a0);
        }
        return a0 + a1;
    }
    // (unrolled: the loads of four steps leave together -- a wide inverse streams from HBM and one
    // wave per row keeps too few bytes in flight otherwise -- the sums are taken in the same order)
#pragma unroll 4
    for (; c + 2 * LPR < k2; c += 4 * LPR) {
        const double2_t r0 = *reinterpret_cast<const double2_t *>(row + c);
        const double2_t r1 = *reinterpret_cast<const double2_t *>(row + c + 2 * LPR);
        const double2_t g0 = *reinterpret_cast<const double2_t *>(ag + c);
        const double2_t g1 = *reinterpret_cast<const double2_t *>(ag + c + 2 * LPR);
        a0 = fma(r0.x, g0.x, a0);
        a1 = fma(r1.x, g1.x, a1);
        a0 = fma(r0.y, g0.y, a0);
        a1 = fma(r1.y, g1.y, a1);
    }
    for (; c < k2; c += 2 * LPR) {
        const double2_t r0 = *reinterpret_cast<const double2_t *>(row + c);
        const double2_t g0 = *reinterpret_cast<const double2_t *>(ag + c);
        a0 = fma(r0.x, g0.x, a0);
        a0 = fma(r0.y, g0.y, a0);
    }
    return a0 + a1;
}

template <int LPR>
__device__ __forceinline__ double fast_gemv_row_tail(double acc, int i, int m, int neta,
                                                     const double *__restrict__ U, long long ldu,
                                                     const double *__restrict__ beta, int sub)
{
    if (i < m)
        for (int t = sub; t < neta; t += LPR) acc = fma(-U[(long long)t * ldu + i], beta[t], acc);
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DZG_WAVE);
    return acc;
}

template <int LPR, int VAR = 0>
__device__ __forceinline__ double fast_gemv_row(int i, int m, int k2, int neta,
                                                const double *__restrict__ binv, long long ldb,
                                                const double *__restrict__ ag,
                                                const double *__restrict__ U, long long ldu,
                                                const double *__restrict__ beta, int sub)
{
    const double acc = fast_gemv_row_head<LPR, VAR>(i, m, k2, binv, ldb, ag, sub);
    return fast_gemv_row_tail<LPR>(acc, i, m, neta, U, ldu, beta, sub);
}

// v_r = (row p of Binv)_r = base - sum_t U_t[p] W_t[r]  (BTRAN): the eta file's share, sixteen etas
// per trip to memory; an eta beyond neta contributes fma(0, 0, acc) = acc exactly.
__device__ __forceinline__ double fast_btran_eta(int neta, const double *__restrict__ U, long long ldu,
                                                 const double *__restrict__ W, long long ldw, int p,
                                                 int r)
{
    double acc = 0.0;
    for (int t0 = 0; t0 < neta; t0 += 16) {
        double u[16], w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool ok = t0 + j < neta;
            u[j] = ok ? U[(long long)(t0 + j) * ldu + p] : 0.0;
            w[j] = ok ? W[(long long)(t0 + j) * ldw + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = fma(u[j], w[j], acc);
    }
    return acc;
}

// ... and the share of the unit columns: position i holds the slack of row rr (bc = its column
// code < 0), whose column of the inverse is e_i: it contributes a_j[rr]
__device__ __forceinline__ double fast_gemv_unit(double acc, int bc, int code,
                                                 const double *__restrict__ acolp)
{
    if (bc < 0) {
        const int rr = -1 - bc;
        acc += code >= 0 ? acolp[rr] : ((-1 - code) == rr ? 1.0 : 0.0);
    }
    return acc;
}

// ---------------------------------------------------------------------------------
// The pivot's books: step lengths and the finiteness assert (src/simplex.rs:257-260,:464-468), swap
// (:239-251), pivot log, then the basis bookkeeping: eta append, compact column append /
// delete, list of nonbasic structural positions.  Work for ONE workgroup; it has no kernel of its
// own.  Seven-launch iteration: it runs in the dual-step launch of k_fast_gemv, which precedes the
// update kernel -- as workgroup 0 in a primal step (that launch has nothing else to do then: dx,
// dz, p and r are all known), as the workgroup that computes row p of dx in a dual step.  Chain:
// inside k_chain_post, by the same two workgroups.  `c` is the control block as the kernel found it.
// ---------------------------------------------------------------------------------
struct DzgPivotArgs {
    int m, q;
    const double *x, *xbar, *z, *zbar, *dx, *dz, *v;
    int *basis, *nonbasis;
    int *bcode, *nbcode;
    const int *var_col;
    double *binv;
    long long ldb;
    int *drow, *dslot;
    double *W;
    long long ldw;
    int *plist, *pslot, *pcode;
    int *cpos;         // code -> nonbasic position (nullptr: not kept)
    int price_small;   // this iteration's pricing pass was the fused small-k row kernel (bytes accounting)
    int col0, col1;
    const long long *cptr;
    int *log_kind, *log_enter, *log_leave;
    double *log_mu, *log_margin;
    long long log_cap;
    int rows_T;        // row-wise pricing while ncompact < rows_T (0: never), rows of rows_ld doubles
    long long rows_ld;
    int row0, row1;    // rows of Binv0 this device holds ([0, m) unless the basis side is row-sharded)
};

inline DzgPivotArgs dzg_pivot_args(const DzgDev &d)
{
    DzgPivotArgs pa;
    pa.m = d.m; pa.q = d.q;
    pa.x = d.x; pa.xbar = d.xbar; pa.z = d.z; pa.zbar = d.zbar; pa.dx = d.dx; pa.dz = d.dz; pa.v = d.v;
    pa.basis = d.basis; pa.nonbasis = d.nonbasis; pa.var_col = d.var_col;
    pa.bcode = d.bcode; pa.nbcode = d.nbcode;
    pa.binv = d.binv; pa.ldb = d.ldb; pa.drow = d.drow; pa.dslot = d.dslot;
    pa.W = d.W; pa.ldw = d.ldw; pa.plist = d.plist; pa.pslot = d.pslot; pa.pcode = d.pcode;
    pa.cpos = d.cpos; pa.price_small = 0;
    pa.col0 = d.col0; pa.col1 = d.col1; pa.cptr = d.csc ? d.cptr : nullptr;
    pa.log_kind = d.log_kind; pa.log_enter = d.log_enter; pa.log_leave = d.log_leave;
    pa.log_mu = d.log_mu; pa.log_margin = d.log_margin; pa.log_cap = d.log_cap;
    pa.rows_T = d.At ? d.rows_T : 0; pa.rows_ld = d.ldt;
    pa.row0 = d.rs ? d.rs_r0 : 0; pa.row1 = d.rs ? d.rs_r1 : d.m;
    return pa;
}

struct DzgPivotScalars {
    double t, s, tbar, sbar; // src/simplex.rs:257-260
    double max_err;          // health monitor after this pivot
    int ok;                  // 0: safe_divide's assert fired (src/simplex.rs:466)
};

__device__ __forceinline__ DzgPivotScalars fast_pivot_scalars(double xp, double xbp, double dxp,
                                                              double zr, double zbr, double dzr,
                                                              int neta, double max_err)
{
    DzgPivotScalars ps;
    int ok = 1;
    ps.t = dzg_safe_divide(xp, dxp, &ok);
    ps.s = dzg_safe_divide(zr, dzr, &ok);
    ps.tbar = dzg_safe_divide(xbp, dxp, &ok);
    ps.sbar = dzg_safe_divide(zbr, dzr, &ok);
    if (neta >= R_) ok = 0; // the host flushes every DZG_RMAX pivots; never reached
    // the pivot element is known twice: dx_p = (B^-1 a_j)_p from FTRAN and -dz_r = v . a_j
    // from BTRAN + pricing.  Their disagreement measures what the explicit inverse lost.
    {
        const double a1 = fabs(dxp), a2 = fabs(dzr);
        const double den = a1 > a2 ? a1 : a2;
        const double err = den > 0.0 ? fabs(dxp + dzr) / den : 0.0;
        if (err > max_err) max_err = err;
    }
    ps.max_err = max_err;
    ps.ok = ok;
    return ps;
}

// `ps` need only be valid in thread 0.  chain != 0 (k_chain_post): ONE WAVE keeps the books (lanes
// `tid` 0..63, no workgroup barrier inside) while the rest of the launch is updating the vectors,
// from their own copies of these scalars; the new eta / column counts are committed here (nobody in
// that launch reads them from the control block), and the 1.0 of a column appended to Binv0 is
// written by the thread that owns row p.
__device__ __forceinline__ void fast_pivot_books_s(DzgCtl *ctl, const DzgCtl &c,
                                                   const DzgPivotArgs &pa,
                                                   const DzgPivotScalars &ps, int chain)
{
    __shared__ int sh_ok, sh_k, sh_ci, sh_cj;
    int s_ok = 0, s_k = 0, s_ci = 0, s_cj = 0; // thread 0's copies; broadcast below
    const int m = pa.m, q = pa.q, col0 = pa.col0, col1 = pa.col1;
    int *basis = pa.basis, *nonbasis = pa.nonbasis, *drow = pa.drow, *dslot = pa.dslot;
    int *plist = pa.plist, *pslot = pa.pslot;
    const int *var_col = pa.var_col;
    double *binv = pa.binv, *W = pa.W;
    const long long ldb = pa.ldb, ldw = pa.ldw, log_cap = pa.log_cap;
    const long long *cptr = pa.cptr;
    int *log_kind = pa.log_kind, *log_enter = pa.log_enter, *log_leave = pa.log_leave;
    double *log_mu = pa.log_mu, *log_margin = pa.log_margin;
    const int tid = chain ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int nthr = chain ? 64 : (int)blockDim.x;
    const int p = c.leave_pos, r = c.enter_pos, neta = c.neta;
    const long long s0 = c.nb_struct;
    // everything addressed by p, r -- issued together, used below
    int vi = 0, vj = 0, idx_r = 0, lastpos = 0, lastcode = 0;
    const double max_err = ps.max_err;
    if (tid == 0) {
        vi = basis[p];
        vj = nonbasis[r];
        idx_r = pslot[r];
        lastpos = s0 > 0 ? plist[s0 - 1] : 0;
        lastcode = s0 > 0 ? pa.pcode[s0 - 1] : 0;
        const int ci = var_col[vi], cj = var_col[vj]; // the two column codes
        if (ps.ok) {
            ctl->t = ps.t;
            ctl->s = ps.s;
            ctl->tbar = ps.tbar;
            ctl->sbar = ps.sbar;
        } else {
            ctl->status = DZG_PANIC; // assert in safe_divide, src/simplex.rs:466
        }
        s_ok = ps.ok;
        s_k = c.ncompact;
        s_ci = ci;
        s_cj = cj;
    }
    if (chain) { // one wave: lane 0's values by shuffle
        s_ok = __shfl(s_ok, 0, DZG_WAVE);
        s_k = __shfl(s_k, 0, DZG_WAVE);
        s_ci = __shfl(s_ci, 0, DZG_WAVE);
        s_cj = __shfl(s_cj, 0, DZG_WAVE);
    } else {
        if (tid == 0) {
            sh_ok = s_ok;
            sh_k = s_k;
            sh_ci = s_ci;
            sh_cj = s_cj;
        }
        __syncthreads();
        s_ok = sh_ok;
        s_k = sh_k;
        s_ci = sh_ci;
        s_cj = sh_cj;
    }
    if (!s_ok) return;
    const int ci = s_ci, cj = s_cj;
    // (the eta of this pivot, u = (dx - e_p)/dx_p and w = v, is appended by the update, which
    // runs on the whole chip: here one workgroup only keeps the books)
    // ---- a leaving slack makes the column of its row dense: it was e_p
    if (ci < 0 && tid == 0) {
        const int k = s_k, rl = -1 - ci;
        drow[k] = rl;
        dslot[rl] = k;
        if (!chain && p >= pa.row0 && p < pa.row1)
            binv[(long long)p * ldb + k] = 1.0; // columns >= ncompact are kept zero
        s_k = k + 1;
    }
    // ---- an entering slack makes the column of its row the unit vector e_p again: its compact
    // column is deleted (swap with the last).  The books are kept here; the m-row column move
    // itself is done by the update on the whole chip (ctl->del_ce / del_last).
    int del_ce = -1, del_last = -1;
    if (cj < 0) {
        const int re = -1 - cj;
        for (int t = tid; t < neta; t += nthr) W[(long long)t * ldw + re] = 0.0;
        if (tid == 0) {
            const int ce = dslot[re], last = s_k - 1;
            if (ce != last) {
                const int lr = drow[last];
                drow[ce] = lr;
                dslot[lr] = ce;
            }
            dslot[re] = -1;
            s_k = last;
            del_ce = ce;
            del_last = last;
        }
    }
    if (tid != 0) return;
    // ---- swap, log, counters (single lane)
    const long long it = c.iter;
    if (it < log_cap) {
        log_kind[it] = c.kind;
        log_enter[it] = vj;
        log_leave[it] = vi;
        log_mu[it] = c.mu;
    }
    long long s = s0;
    // algorithmic bytes of this iteration's pricing pass; sparse: 12 B per stored
    // entry of the nonbasic structural columns + their column pointers
    double bytes = c.price_bytes;
    if (!cptr) ctl->price_mask = c.price_mask | ((pa.rows_T > 0 && c.ncompact < pa.rows_T) ? 1 : 2);
    if (cptr)
        bytes += 12.0 * (double)c.nb_nnz + 4.0 * (double)(s + 1) + 8.0 * (double)m + 32.0 * (double)q;
    else if (pa.rows_T > 0 && c.ncompact < pa.rows_T) {
        // row-wise pass: k (+ 1) rows of the row-major copy, the groups' partial sums written and
        // read once, v's k coefficients, and the per-position part as below
        const double nrows = (double)c.ncompact + (ci < 0 ? 1.0 : 0.0); // (a leaving slack's own row)
        double G = ceil(((double)c.ncompact + 1.0) / 16.0);
        G = G > 32.0 ? 32.0 : G;
        // (the fused small-k kernel keeps the partial sums in LDS and reads the code -> position map)
        bytes += 8.0 * nrows * (double)pa.rows_ld +
                 (pa.price_small ? 4.0 * (double)(col1 - col0) : 16.0 * G * (double)pa.rows_ld) + 12.0 * nrows +
                 32.0 * (double)q;
    } else
        bytes += 8.0 * (double)m * (double)s + 8.0 * (double)m + 32.0 * (double)q;
    ctl->price_bytes = bytes;
    basis[p] = vj;
    nonbasis[r] = vi;
    pa.bcode[p] = cj;
    pa.nbcode[r] = ci;
    // nonbasic position r now holds vi instead of vj; the list only tracks OWNED columns
    const bool own_j = cj >= col0 && cj < col1, own_i = ci >= col0 && ci < col1;
    if (cptr) {
        long long nnz = c.nb_nnz;
        if (own_j) nnz -= cptr[cj - col0 + 1] - cptr[cj - col0];
        if (own_i) nnz += cptr[ci - col0 + 1] - cptr[ci - col0];
        ctl->nb_nnz = nnz;
    }
    if (pa.cpos) { // column code -> nonbasic position (k_price_rows_small)
        if (own_j) pa.cpos[cj - col0] = -1;
        if (own_i) pa.cpos[ci - col0] = r;
    }
    if (own_j && !own_i) { // an owned structural column left the nonbasic set
        plist[idx_r] = lastpos;
        pa.pcode[idx_r] = lastcode;
        pslot[lastpos] = idx_r;
        pslot[r] = -1;
        --s;
    } else if (!own_j && own_i) {
        plist[s] = r;
        pa.pcode[s] = ci;
        pslot[r] = (int)s;
        ++s;
    } else if (own_j && own_i) { // position r stays in the list with another column
        pa.pcode[idx_r] = ci;
    }
    ctl->nb_struct = s;
    ctl->enter_var = vj;
    ctl->leave_var = vi;
    // seven launches: committed by k_fast_update, since this launch still reads the old counts
    ctl->ncompact_next = s_k;
    ctl->del_ce = del_ce;
    ctl->del_last = del_last;
    ctl->neta_next = neta + 1;
    if (chain) {
        ctl->ncompact = s_k;
        ctl->neta = neta + 1;
    }
    ctl->max_pivot_err = max_err;
    // near-tie record of this pivot; the tolerance follows the health monitor
    if (it < log_cap) log_margin[it] = c.margin;
    if (c.margin < c.min_margin) ctl->min_margin = c.margin;
    if (c.tie_seen) {
        ctl->near_ties = c.near_ties + 1;
        if (c.first_near_tie < 0) ctl->first_near_tie = it;
    }
    if (c.tie_tol >= 0.0) {
        double adaptive = 64.0 * max_err;
        if (c.drift_tau > adaptive) adaptive = c.drift_tau;
        ctl->tau = adaptive > c.tie_tol ? adaptive : c.tie_tol;
    }
    ctl->iter = it + 1;
}

// the seven-launch form: thread 0 reads the six values the step lengths need
__device__ __forceinline__ void fast_pivot_books(DzgCtl *ctl, const DzgCtl &c, const DzgPivotArgs &pa,
                                                 double dxp)
{
    DzgPivotScalars ps;
    ps.ok = 1;
    if (threadIdx.x == 0) {
        const int p = c.leave_pos, r = c.enter_pos;
        const bool rec = c.use_record != 0;
        const double xp = pa.x[p], xbp = pa.xbar[p]; // (dx_p comes from the workgroup's own GEMV row)
        const double zr = rec ? c.zr : pa.z[r], zbr = rec ? c.zbar_r : pa.zbar[r];
        const double dzr = rec ? c.dz_r : pa.dz[r];
        ps = fast_pivot_scalars(xp, xbp, dxp, zr, zbr, dzr, c.neta, c.max_pivot_err);
    }
    fast_pivot_books_s(ctl, c, pa, ps, 0);
}

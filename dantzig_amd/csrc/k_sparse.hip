// k_sparse.hip -- FAST numerics for a constraint matrix kept sparse (CSC) on the device: the
// basis is represented on its k x k structural block only, and FTRAN / BTRAN cost what the
// sparse right-hand sides and the matrix's nonzeros cost, not 8*m*k bytes.
//
// The reference densifies everything (src/linalg.rs:236-238, src/simplex.rs:228,234) and has no
// sparse basis to follow; this is SURVEY 8(f4).  With B = [A_S | E_L] (S: the k structural basic
// columns, L: rows whose slack is basic, R: the other k rows) only
//
//      X = (A[R, S])^-1          k x k, dense, product form  X - Ub^T Wc  (<= 64 pending etas)
//
// is stored (the dense-inverse path of k_fast.hip stores all m rows: m x k).  Everything else
// follows from the rows of  B dx = a  and the columns of  B^T v = e_p  that belong to slacks:
//
//   FTRAN  dx_S = X a_R                      a_R: the <= nnz(a_j) entries of a_j in rows of R, so
//                                            only those COLUMNS of X are read (k * nnz(a_R) values)
//          dx_p' = a_j[r'] - A[r', S] dx_S   for the basic slack of row r' (position p'): one pass
//                                            over a CSR copy of A, 12 bytes per stored entry,
//                                            deterministic order (no atomics)
//   BTRAN  v_R = row p of X                  p structural;  or  v_R = -A[r', S] X, v[r'] = 1
//                                            p the slack of row r': only the rows of X whose
//                                            column has an entry in row r' are combined
//   update eta append (u on the structural positions, w = v_R); a leaving slack appends a row
//          and a column, an entering slack deletes one of each (the last moves into the hole);
//          every 64 pivots X -= Ub^T Wc on the fp64 matrix cores.
//
// Memory: 8 k^2 bytes (20 GB at k = m = 50 000, BASELINE config 4; the refactorisation
// workspace, reserved on first need, is twice that again).  A factorisation of A[R, S] in sparse
// form would not be smaller on this family: a uniformly random pattern has no structure to
// preserve, and its LU factors fill in to near-dense beyond a few thousand columns (DESIGN.md
// section 4 quotes the fill measured with SuperLU).  What the sparse structure does buy -- and
// what this path uses -- is the sparsity of the right-hand sides.
#include "common.h"
#include "fast_decide.h"
#include "chain_barrier.h"

#define R_ DZG_RMAX
#define SP_NB 1024 // workgroups of the m-sized kernels at most (fixed fan-in of their partials)
#define SP_NB_UPD 256 // workgroups of k_sp_update = first-pivot partials per side

__device__ __forceinline__ double sp_block_sum(double x)
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

// ---------------------------------------------------------------------------------
// k_sp_ftran_s<KIND>: the head of a step and dx on the structural basis positions.
//   head (every workgroup, redundantly, from the same partial arrays: no preparation launch):
//     KIND = PRIMAL  status() at the head of the iteration (src/simplex.rs:274-306); a dual step
//                    ends the launch here;
//     KIND = DUAL    the dual step's ratio test after pricing (:324-325);
//     then the entering column's entries that lie in rows of R go through LDS and
//     beta_t = W_t . a_R (a sparse dot, thread t) is formed; workgroup 0 also swaps the dense
//     copy of the entering column in `acol` (the previous column's entries are cleared, so acol
//     is zero outside the current column without an O(m) pass) and publishes the decision.
//   body: dx_S[b] = sum_e X[b][slot_e] a_e - sum_t Ub[t][b] beta_t, one thread per row b of X.
// A primal step leaves per-workgroup ratio-test candidates (src/simplex.rs:439-461).
// grid = min(ceil(m / 256), SP_NB) workgroups of 256 (k <= m is only known on the device).
// ---------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_sp_ftran_s(
    DzgCtl *ctl, int m, const long long *__restrict__ cptr, const int *__restrict__ ridx,
    const double *__restrict__ cval, const int *__restrict__ nbcode,
    const double *__restrict__ fpx_r,
    const int *__restrict__ fpx_k, const double *__restrict__ fpx_h,
    const double *__restrict__ fpz_r, const int *__restrict__ fpz_k,
    const double *__restrict__ fpz_h, const double *__restrict__ rz_r,
    const int *__restrict__ rz_k, const double *__restrict__ rz_h, int nrz,
    const double *__restrict__ X, long long ldb, const double *__restrict__ U, long long ldu,
    const double *__restrict__ W, long long ldw, const int *__restrict__ dslot,
    const int *__restrict__ spos, const double *__restrict__ x, const double *__restrict__ xbar,
    double *__restrict__ dxs, double *__restrict__ dx, double *__restrict__ rx_r,
    int *__restrict__ rx_k, double *__restrict__ rx_h, double *__restrict__ acol, int *acol_code,
    double eps)
{
    __shared__ int s_slot[256];
    __shared__ double s_val[256];
    __shared__ double s_beta[R_];
    __shared__ int s_wcnt[4];
    // (the partial candidates are fetched beside the control block, not behind it: after a kernel
    // boundary every first touch is a trip to memory, and these two need not queue)
    DzgCand2 cj = dzg_cand2_none(), ci = dzg_cand2_none(), cw = dzg_cand2_none();
    if (KIND == DZG_STEP_PRIMAL) {
        cj = reduce_partials(fpz_r, fpz_k, fpz_h, SP_NB_UPD);
        ci = reduce_partials(fpx_r, fpx_k, fpx_h, SP_NB_UPD);
    } else {
        cw = reduce_partials(rz_r, rz_k, rz_h, nrz);
    }
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    // ---- head: every workgroup takes the decision itself (the control block may already carry
    // the lead's version of it: only fields no decision writes are read from the snapshot)
    int epos;
    double mu;
    if (KIND == DZG_STEP_PRIMAL) {
        int kind;
        if (!fast_status(ctl, c, lead, cj, ci, eps, m, false, kind, &mu)) return;
        if (kind != DZG_STEP_PRIMAL) return;
        epos = cj.k;
    } else {
        if (c.kind != DZG_STEP_DUAL) return; // (written by the primal launch of this iteration)
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) return;
        epos = cw.k;
        mu = c.mu;
        if (lead) ctl->enter_pos = epos;
    }
    const int code = nbcode[epos];
    if (lead) ctl->enter_code = code;
    const int k = c.ncompact, neta = c.neta;
    const double tau = c.tau;
    const long long e0 = code >= 0 ? cptr[code] : 0, e1 = code >= 0 ? cptr[code + 1] : 1;
    if (blockIdx.x == 0) { // the dense copy of the entering column, for k_sp_ftran_l
        const int prev = *acol_code; // INT_MIN: nothing scattered yet
        if (prev != (int)0x80000000) {
            if (prev < 0) {
                if (threadIdx.x == 0) acol[-1 - prev] = 0.0;
            } else {
                for (long long e = cptr[prev] + threadIdx.x; e < cptr[prev + 1]; e += blockDim.x)
                    acol[ridx[e]] = 0.0;
            }
        }
        __syncthreads(); // the two columns may share rows
        if (code < 0) {
            if (threadIdx.x == 0) acol[-1 - code] = 1.0;
        } else {
            for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) acol[ridx[e]] = cval[e];
        }
        if (threadIdx.x == 0) *acol_code = code;
    }
    // ---- beta_t = W_t . a_R, thread t, over the column's entries in chunks of 256.  A chunk is
    // staged COMPACTED -- only the entries in rows of R (dslot >= 0), in their order -- so that the
    // gathers below are loops without branches whose loads leave four at a time: a taken branch
    // around every load made each entry of a_R a trip to memory of its own (1 at k = 1 000, 45
    // deep in the solve of config 4).  The sums are the same sums in the same order.
    auto stage = [&](long long base) -> int {
        __syncthreads(); // the previous chunk has been consumed
        const long long e = base + threadIdx.x;
        int slot = -1;
        double val = 0.0;
        if (e < e1) {
            const int r = code >= 0 ? ridx[e] : -1 - code;
            slot = dslot[r];
            val = code >= 0 ? cval[e] : 1.0;
        }
        const unsigned long long mask = __ballot(slot >= 0);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        if (slot >= 0) {
            const int at = off + __popcll(mask & ((1ull << lane) - 1ull));
            s_slot[at] = slot;
            s_val[at] = val;
        }
        const int total = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
        return total;
    };
    double bacc = 0.0;
    int cnt = 0;
    for (long long base = e0; base < e1; base += 256) {
        cnt = stage(base);
        if ((int)threadIdx.x < neta) {
            const double *wt = W + (long long)threadIdx.x * ldw;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double w0 = wt[s_slot[i]], w1 = wt[s_slot[i + 1]], w2 = wt[s_slot[i + 2]],
                             w3 = wt[s_slot[i + 3]];
                bacc = fma(w0, s_val[i], bacc);
                bacc = fma(w1, s_val[i + 1], bacc);
                bacc = fma(w2, s_val[i + 2], bacc);
                bacc = fma(w3, s_val[i + 3], bacc);
            }
            for (; i < cnt; ++i) bacc = fma(wt[s_slot[i]], s_val[i], bacc);
        }
    }
    if (threadIdx.x < R_) s_beta[threadIdx.x] = (int)threadIdx.x < neta ? bacc : 0.0;
    __syncthreads(); // (a column without stored entries runs none of the staging barriers)
    // ---- body (the last chunk of the column is still staged when there is only one)
    DzgCand2 best = dzg_cand2_none();
    const bool one_chunk = e1 - e0 <= 256;
    for (int b0 = blockIdx.x * blockDim.x; b0 < k; b0 += gridDim.x * blockDim.x) { // block-uniform
        const int b = b0 + threadIdx.x;
        const double *row = X + (long long)(b < k ? b : 0) * ldb;
        // what does not depend on the staged column leaves first: the position, its x and xbar
        const int i = b < k ? spos[b] : 0;
        double xi = 0.0, xbi = 0.0;
        if (KIND == DZG_STEP_PRIMAL && b < k) {
            xi = x[i];
            xbi = xbar[i];
        }
        double acc = 0.0;
        for (long long base = e0; base < e1; base += 256) {
            if (!one_chunk) cnt = stage(base);
            else __syncthreads(); // (orders s_beta before its first use)
            if (b < k) {
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const double r0 = row[s_slot[j]], r1 = row[s_slot[j + 1]], r2 = row[s_slot[j + 2]],
                                 r3 = row[s_slot[j + 3]];
                    acc = fma(r0, s_val[j], acc);
                    acc = fma(r1, s_val[j + 1], acc);
                    acc = fma(r2, s_val[j + 2], acc);
                    acc = fma(r3, s_val[j + 3], acc);
                }
                for (; j < cnt; ++j) acc = fma(row[s_slot[j]], s_val[j], acc);
            }
        }
        if (b < k) {
            int t = 0;
            for (; t + 8 <= neta; t += 8) { // (eight coalesced loads side by side, the sum in order)
                double u[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) u[g] = U[(long long)(t + g) * ldu + b];
#pragma unroll
                for (int g = 0; g < 8; ++g) acc = fma(-u[g], s_beta[t + g], acc);
            }
            for (; t < neta; ++t) acc = fma(-U[(long long)t * ldu + b], s_beta[t], acc);
            dxs[b] = acc;
            dx[i] = acc;
            if (KIND == DZG_STEP_PRIMAL) {
                const double scaled = mu * xbi;
                const double den = xi + scaled;
                DzgCand2 cnd;
                cnd.r = dzg_div(acc, den);
                cnd.k = i;
                cnd.h = -__builtin_inf();
                if (cnd.r > 0.0) best = dzg_better2(best, cnd);
                if (dzg_noise_zero(den, xi, scaled, tau)) best.h = __builtin_inf();
            }
        }
    }
    if (KIND == DZG_STEP_PRIMAL) {
        best = dzg_block_best2(best);
        if (threadIdx.x == 0) {
            rx_r[blockIdx.x] = best.r;
            rx_k[blockIdx.x] = best.k;
            rx_h[blockIdx.x] = best.h;
        }
    }
}

// ---------------------------------------------------------------------------------
// k_sp_ftran_l: dx on the positions of the basic slacks, from the rows of B dx = a_j:
//     dx[p'] = a_j[r'] - sum_{col basic, A[r', col] stored} A[r', col] * dx_S[row of X of col]
// Each constraint row keeps the list of its entries in BASIC structural columns (bcnt / bcol /
// bval, in the row's slice of the CSR-shaped buffers; k_sp_pivot appends the entering column's
// entries and removes the leaving one's), so the pass touches k * nnz-per-column entries, not
// nnz(A).  One thread per row walks its list in list order: deterministic, no atomics.
// grid = min(ceil(m / 256), SP_NB) workgroups of 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sp_ftran_l(
    const DzgCtl *ctl, int need_kind, int m, const long long *__restrict__ rptr,
    const int *__restrict__ bcnt, const int *__restrict__ bcol, const double *__restrict__ bval,
    const int *__restrict__ bslot, const int *__restrict__ rowpos,
    const double *__restrict__ acol, const double *__restrict__ dxs,
    const double *__restrict__ x, const double *__restrict__ xbar, double *__restrict__ dx,
    double *__restrict__ rx_r, int *__restrict__ rx_k, double *__restrict__ rx_h, int part0)
{
    // this thread's first row: what only depends on the row leaves beside the control block
    const int r_first = blockIdx.x * blockDim.x + threadIdx.x;
    int p_first = -1, n_first = 0;
    double a_first = 0.0;
    long long e_first = 0;
    if (r_first < m) {
        p_first = rowpos[r_first];
        a_first = acol[r_first];
        e_first = rptr[r_first];
        n_first = bcnt[r_first];
    }
    const DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING || c.kind != need_kind) return;
    const double mu = c.mu, tau = c.tau;
    DzgCand2 best = dzg_cand2_none();
    for (int r = r_first; r < m; r += gridDim.x * blockDim.x) {
        const bool first = r == r_first;
        const int p = first ? p_first : rowpos[r]; // -1: the slack of row r is nonbasic (r in R)
        if (p < 0) continue;
        double acc = first ? a_first : acol[r];
        const long long e0 = first ? e_first : rptr[r];
        const int n = first ? n_first : bcnt[r];
        double xi = 0.0, xbi = 0.0;
        if (need_kind == DZG_STEP_PRIMAL) {
            xi = x[p];
            xbi = xbar[p];
        }
        int i = 0;
        for (; i + 4 <= n; i += 4) { // four entries' chains (column -> row of X -> dx_S) side by side
            const int c0 = bcol[e0 + i], c1 = bcol[e0 + i + 1], c2 = bcol[e0 + i + 2], c3 = bcol[e0 + i + 3];
            const double v0 = bval[e0 + i], v1 = bval[e0 + i + 1], v2 = bval[e0 + i + 2], v3 = bval[e0 + i + 3];
            const int s0 = bslot[c0], s1 = bslot[c1], s2 = bslot[c2], s3 = bslot[c3];
            const double d0 = dxs[s0], d1 = dxs[s1], d2 = dxs[s2], d3 = dxs[s3];
            acc = fma(-v0, d0, acc);
            acc = fma(-v1, d1, acc);
            acc = fma(-v2, d2, acc);
            acc = fma(-v3, d3, acc);
        }
        for (; i < n; ++i) acc = fma(-bval[e0 + i], dxs[bslot[bcol[e0 + i]]], acc);
        dx[p] = acc;
        if (need_kind == DZG_STEP_PRIMAL) {
            const double scaled = mu * xbi;
            const double den = xi + scaled;
            DzgCand2 cnd;
            cnd.r = dzg_div(acc, den);
            cnd.k = p;
            cnd.h = -__builtin_inf();
            if (cnd.r > 0.0) best = dzg_better2(best, cnd);
            if (dzg_noise_zero(den, xi, scaled, tau)) best.h = __builtin_inf();
        }
    }
    if (need_kind == DZG_STEP_PRIMAL) {
        best = dzg_block_best2(best);
        if (threadIdx.x == 0) {
            rx_r[part0 + blockIdx.x] = best.r;
            rx_k[part0 + blockIdx.x] = best.k;
            rx_h[part0 + blockIdx.x] = best.h;
        }
    }
}

// ---------------------------------------------------------------------------------
// k_sp_btran: v = row p of B^-1 in row coordinates.  A primal step first finishes its ratio
// test (none = Unbounded, src/simplex.rs:313).  The row is a sparse combination of rows of X:
//   p structural (row b of X):     L = {(b, 1)}
//   p the basic slack of row r':   L = {(row of X of col, -A[r', col]) : col basic},  v[r'] = 1
// v_R[c] = sum_L coef * X[b][c] - sum_t (sum_L coef * Ub[t][b]) * Wc[t][c];  v = 0 elsewhere.
// Every workgroup reads L (the row's list of basic entries) and forms the 64 gammas, then fills
// its share of v.
// grid = min(ceil(m / 256), SP_NB) workgroups of 256.
// ---------------------------------------------------------------------------------
#define SP_LCAP 1024
__global__ __launch_bounds__(256) void k_sp_btran(
    DzgCtl *ctl, int m, int nparts, const long long *__restrict__ rptr,
    const int *__restrict__ bcnt, const int *__restrict__ bcol, const double *__restrict__ bval,
    const int *__restrict__ bslot, const int *__restrict__ sslot, const int *__restrict__ bcode,
    const double *__restrict__ X, long long ldb,
    const double *__restrict__ U, long long ldu, const double *__restrict__ W, long long ldw,
    const int *__restrict__ drow, const int *__restrict__ dslot,
    const double *__restrict__ rx_r, const int *__restrict__ rx_k,
    const double *__restrict__ rx_h, double *__restrict__ v, const long long *__restrict__ cptr,
    const int *__restrict__ cidx, const double *__restrict__ rval, int *lcnt, DzgLiveEntry *lent)
{
    __shared__ int s_b[SP_LCAP];
    __shared__ double s_coef[SP_LCAP];
    __shared__ double s_gamma[R_];
    __shared__ int s_cnt;
    // (a primal step's ratio partials are fetched beside the control block; a dual step does not
    // use them)
    const DzgCand2 cw = reduce_partials(rx_r, rx_k, rx_h, nparts);
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    int p;
    if (c.kind == DZG_STEP_PRIMAL) {
        const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_UNBOUNDED)) return;
        p = cw.k;
        if (lead) ctl->leave_pos = p;
    } else {
        p = c.leave_pos;
    }
    const int k = c.ncompact, neta = c.neta;
    const int bp = sslot[p];
    const int rl = bp >= 0 ? -1 : -1 - bcode[p]; // row whose slack is basic at p
    const int tid = threadIdx.x, stride = gridDim.x * blockDim.x;
    const int gid = blockIdx.x * blockDim.x + tid;
    // live-entry lists of the columns (k_price_csc_rl): the leaving slack's row carries v = 1 in
    // the pricing pass that follows and joins R at this pivot: its entries join their columns'
    // lists here (a row has at most one entry per column: no two threads share a list)
    // The append is idempotent across a stop: a dual step's ratio test comes AFTER the pricing pass
    // and may end the run before the pivot (DZG_NEAR_TIE, resumable); ctl->rl_listed remembers the
    // row that is already listed, the resumed iteration -- the same decision from the same state --
    // finds it and appends nothing, k_sp_pivot clears the mark when the row has joined R for good.
    // (A pending row that is NOT this iteration's -- unreachable today -- is taken out first.)
    if (lcnt && blockIdx.x == 0 && c.rl_listed != rl) { // (block-uniform: c is a snapshot)
        const int pend = c.rl_listed;
        if (pend >= 0) {
            for (long long e = rptr[pend] + tid; e < rptr[pend + 1]; e += blockDim.x) {
                const int col = cidx[e];
                const long long base = cptr[col];
                const int n = lcnt[col];
                for (int i = 0; i < n; ++i)
                    if (lent[base + i].row == pend) {
                        lent[base + i] = lent[base + n - 1];
                        lcnt[col] = n - 1;
                        break;
                    }
            }
            __syncthreads(); // a column may hold entries of both rows
        }
        if (rl >= 0)
            for (long long e = rptr[rl] + tid; e < rptr[rl + 1]; e += blockDim.x) {
                const int col = cidx[e];
                const long long at = cptr[col] + lcnt[col];
                DzgLiveEntry en;
                en.row = rl;
                en.pad_ = 0;
                en.val = rval[e];
                lent[at] = en;
                lcnt[col] += 1;
            }
        if (tid == 0) ctl->rl_listed = rl;
    }
    // rows outside R: zero, except the leaving slack's own row
    for (int r = gid; r < m; r += stride)
        if (dslot[r] < 0) v[r] = (r == rl) ? 1.0 : 0.0;
    if (k == 0) return;
    // compact columns: accumulate over L in chunks that fit LDS (a dense row of a user model can
    // hold thousands of basic columns)
    const long long e0 = bp >= 0 ? 0 : rptr[rl], e1 = bp >= 0 ? 1 : e0 + bcnt[rl];
    double acc[4] = {0.0, 0.0, 0.0, 0.0}; // this thread's columns gid, gid + stride, ...
    if (tid < R_) s_gamma[tid] = 0.0;
    for (long long base = e0; base < e1; base += SP_LCAP) {
        __syncthreads();
        // ordered compaction of the chunk's basic entries (the order of L fixes the rounding):
        // 256 entries at a time, ballot + popcount ranks keep the CSR order
        int total = 0;
        if (bp >= 0) {
            if (tid == 0) {
                s_b[0] = bp;
                s_coef[0] = 1.0;
            }
            total = 1;
        } else { // the row's basic entries, in list order
            const long long lim = (e1 - base) < SP_LCAP ? (e1 - base) : SP_LCAP;
            for (long long i = tid; i < lim; i += blockDim.x) {
                s_b[i] = bslot[bcol[base + i]];
                s_coef[i] = -bval[base + i];
            }
            total = (int)lim;
        }
        if (tid == 0) s_cnt = total;
        __syncthreads();
        const int cnt = s_cnt;
        if (tid < neta) { // (gathers four at a time, the sums in list order)
            double g = s_gamma[tid];
            const double *ut = U + (long long)tid * ldu;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double u0 = ut[s_b[i]], u1 = ut[s_b[i + 1]], u2 = ut[s_b[i + 2]], u3 = ut[s_b[i + 3]];
                g = fma(s_coef[i], u0, g);
                g = fma(s_coef[i + 1], u1, g);
                g = fma(s_coef[i + 2], u2, g);
                g = fma(s_coef[i + 3], u3, g);
            }
            for (; i < cnt; ++i) g = fma(s_coef[i], ut[s_b[i]], g);
            s_gamma[tid] = g;
        }
        int slot = 0;
        for (int cc = gid; cc < k && slot < 4; cc += stride, ++slot) {
            double a = acc[slot];
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double x0 = X[(long long)s_b[i] * ldb + cc], x1 = X[(long long)s_b[i + 1] * ldb + cc],
                             x2 = X[(long long)s_b[i + 2] * ldb + cc], x3 = X[(long long)s_b[i + 3] * ldb + cc];
                a = fma(s_coef[i], x0, a);
                a = fma(s_coef[i + 1], x1, a);
                a = fma(s_coef[i + 2], x2, a);
                a = fma(s_coef[i + 3], x3, a);
            }
            for (; i < cnt; ++i) a = fma(s_coef[i], X[(long long)s_b[i] * ldb + cc], a);
            acc[slot] = a;
        }
    }
    __syncthreads();
    int slot = 0;
    for (int cc = gid; cc < k; cc += stride, ++slot) {
        double a;
        if (slot < 4) {
            a = acc[slot];
        } else { // more than 4 columns per thread (k > 4 * grid threads = 1 M): not reachable
            a = 0.0;
        }
        const int vr = drow[cc];
        int t = 0;
        for (; t + 8 <= neta; t += 8) { // (eight coalesced loads side by side, the sum in order)
            double w[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) w[g] = W[(long long)(t + g) * ldw + cc];
#pragma unroll
            for (int g = 0; g < 8; ++g) a = fma(-s_gamma[t + g], w[g], a);
        }
        for (; t < neta; ++t) a = fma(-s_gamma[t], W[(long long)t * ldw + cc], a);
        v[vr] = a;
    }
}

// The single-lane part of k_sp_pivot: step lengths, the books of the k x k block, swap, log,
// counters.  Returns 0 when a step length is not finite (DZG_PANIC).
__device__ __forceinline__ int sp_pivot_books(
    DzgCtl *ctl, const DzgCtl &c, int m, int q, int p, int r, int neta, int vi, int vj, int ci,
    int cj, const double *__restrict__ x, const double *__restrict__ xbar,
    const double *__restrict__ z, const double *__restrict__ zbar, const double *__restrict__ dx,
    const double *__restrict__ dz, int *basis, int *nonbasis, const int *__restrict__ var_col,
    int *drow, int *dslot, int *sslot, int *spos, int *bslot, int *rowpos, int *plist, int *pslot,
    const long long *__restrict__ cptr, int *log_kind, int *log_enter, int *log_leave,
    double *log_mu, double *log_margin, long long log_cap, int *bcode, int *nbcode, int *pcode,
    bool live_lists, double dxp)
{
    const double xp = x[p], xbp = xbar[p];
    const double zr = z[r], zbr = zbar[r], dzr = dz[r];
    int ok = 1;
    const double t = dzg_safe_divide(xp, dxp, &ok);
    const double s = dzg_safe_divide(zr, dzr, &ok);
    const double tbar = dzg_safe_divide(xbp, dxp, &ok);
    const double sbar = dzg_safe_divide(zbr, dzr, &ok);
    if (neta >= R_) ok = 0; // the host flushes every DZG_RMAX pivots; never reached
    double max_err = c.max_pivot_err;
    {
        const double a1 = fabs(dxp), a2 = fabs(dzr);
        const double den = a1 > a2 ? a1 : a2;
        const double err = den > 0.0 ? fabs(dxp + dzr) / den : 0.0;
        if (err > max_err) max_err = err;
    }
    if (!ok) {
        ctl->status = DZG_PANIC; // assert in safe_divide, src/simplex.rs:466
        return 0;
    }
    ctl->t = t;
    ctl->s = s;
    ctl->tbar = tbar;
    ctl->sbar = sbar;
    // ---- the k x k block.  k_sp_update reads these with the OLD k (sp_k) and the new one.
    int k = c.ncompact;
    int app = -1, mrow = -1, mcol = -1, zcol = -1;
    const int last = k - 1;
    if (ci >= 0 && cj >= 0) {          // structural for structural: the row of p stays
        bslot[cj] = bslot[ci];
        bslot[ci] = -1;
    } else if (ci < 0 && cj >= 0) {    // a slack leaves: its row becomes a column of X, p a row
        const int rl = -1 - ci;
        drow[k] = rl;
        dslot[rl] = k;
        rowpos[rl] = -1;
        sslot[p] = k;
        spos[k] = p;
        bslot[cj] = k;
        app = k;
        k += 1;
    } else if (ci >= 0 && cj < 0) {    // a slack enters: row of p and column of its row go
        const int re = -1 - cj, bp = sslot[p], ce = dslot[re];
        bslot[ci] = -1;
        if (bp != last) {
            const int pl = spos[last];
            spos[bp] = pl;
            sslot[pl] = bp;
            bslot[var_col[basis[pl]]] = bp;
            mrow = bp;
        }
        sslot[p] = -1;
        if (ce != last) {
            const int lr = drow[last];
            drow[ce] = lr;
            dslot[lr] = ce;
            mcol = ce;
        }
        dslot[re] = -1;
        rowpos[re] = p;
        k -= 1;
    } else {                           // slack for slack: the column slot changes hands
        const int rl = -1 - ci, re = -1 - cj, ce = dslot[re];
        drow[ce] = rl;
        dslot[rl] = ce;
        dslot[re] = -1;
        rowpos[rl] = -1;
        rowpos[re] = p;
        zcol = ce;
    }
    ctl->sp_k = c.ncompact;
    ctl->sp_app = app;
    ctl->sp_mrow = mrow;
    ctl->sp_mcol = mcol;
    ctl->sp_zcol = zcol;
    ctl->ncompact = k;
    ctl->rl_listed = -1; // (the row k_sp_btran listed ahead of this pivot is a row of R now)
    // ---- swap, log, counters, list of nonbasic structural positions (as fast_pivot_books of k_fast.hip)
    const long long it = c.iter;
    if (it < log_cap) {
        log_kind[it] = c.kind;
        log_enter[it] = vj;
        log_leave[it] = vi;
        log_mu[it] = c.mu;
        log_margin[it] = c.margin;
    }
    long long ns = c.nb_struct;
    // (live-entry pricing: the 16 bytes per walked entry are counted by the kernel itself, rl_work)
    ctl->price_bytes = c.price_bytes +
                       (live_lists ? 20.0 * (double)ns
                                   : 12.0 * (double)c.nb_nnz + 4.0 * (double)(ns + 1)) +
                       8.0 * (double)m + 32.0 * (double)q;
    basis[p] = vj;
    nonbasis[r] = vi;
    bcode[p] = cj;
    nbcode[r] = ci;
    long long nnz = c.nb_nnz;
    if (cj >= 0) nnz -= cptr[cj + 1] - cptr[cj];
    if (ci >= 0) nnz += cptr[ci + 1] - cptr[ci];
    ctl->nb_nnz = nnz;
    if (cj >= 0 && ci < 0) { // a structural column left the nonbasic set
        const int idx = pslot[r], lastpos = plist[ns - 1], lastcode = pcode[ns - 1];
        plist[idx] = lastpos;
        pcode[idx] = lastcode;
        pslot[lastpos] = idx;
        pslot[r] = -1;
        --ns;
    } else if (cj < 0 && ci >= 0) {
        plist[ns] = r;
        pcode[ns] = ci;
        pslot[r] = (int)ns;
        ++ns;
    } else if (cj >= 0 && ci >= 0) { // position r stays in the list with another column
        pcode[pslot[r]] = ci;
    }
    ctl->nb_struct = ns;
    ctl->enter_var = vj;
    ctl->leave_var = vi;
    ctl->neta = neta + 1;
    ctl->max_pivot_err = max_err;
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
    return 1;
}

// ---------------------------------------------------------------------------------
// k_sp_pivot: step lengths and the finiteness assert (src/simplex.rs:257-260,:464-468), swap
// (:239-251), pivot log, and the books of the k x k block: which rows / columns of X appear,
// disappear or are recycled.  The data moves themselves are k_sp_update's (whole chip); they are
// described by ctl->sp_*.  One workgroup.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sp_pivot(
    DzgCtl *ctl, int m, int q, const double *__restrict__ x, const double *__restrict__ xbar,
    const double *__restrict__ z, const double *__restrict__ zbar, const double *__restrict__ dx,
    const double *__restrict__ dz, int *basis, int *nonbasis, const int *__restrict__ var_col,
    int *drow, int *dslot, int *sslot, int *spos, int *bslot, int *rowpos, int *plist, int *pslot,
    const long long *__restrict__ cptr, const int *__restrict__ ridx,
    const double *__restrict__ cval, const long long *__restrict__ rptr, int *bcnt, int *bcol,
    double *bval, int *log_kind, int *log_enter, int *log_leave, double *log_mu,
    double *log_margin, long long log_cap, int *bcode, int *nbcode, int *pcode,
    const int *__restrict__ cidx, int *lcnt, DzgLiveEntry *lent)
{
    __shared__ int s_ok, s_ci, s_cj;
    const DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    if (threadIdx.x == 0) { // one lane reads the pivot's variables BEFORE it swaps them
        const int p = c.leave_pos, r = c.enter_pos;
        const int vi = basis[p], vj = nonbasis[r];
        s_ci = bcode[p]; // (= var_col[vi], in the same trip to memory as vi)
        s_cj = nbcode[r];
        s_ok = sp_pivot_books(ctl, c, m, q, p, r, c.neta, vi, vj, s_ci, s_cj, x, xbar, z, zbar, dx, dz,
                              basis, nonbasis, var_col, drow, dslot, sslot, spos, bslot, rowpos,
                              plist, pslot, cptr, log_kind, log_enter, log_leave, log_mu,
                              log_margin, log_cap, bcode, nbcode, pcode, lcnt != nullptr, dx[p]);
    }
    __syncthreads();
    if (!s_ok) return;
    const int ci = s_ci, cj = s_cj;
    // per-row lists of entries in basic columns: drop the leaving column's, append the entering
    // one's.  A column has at most one entry per row: no two threads touch the same list within
    // a phase; the barrier orders removal before insertion for rows both columns touch.
    if (ci >= 0)
        for (long long e = cptr[ci] + threadIdx.x; e < cptr[ci + 1]; e += blockDim.x) {
            const int row = ridx[e];
            const long long e0 = rptr[row];
            const int n = bcnt[row];
            for (int i = 0; i < n; ++i)
                if (bcol[e0 + i] == ci) {
                    bcol[e0 + i] = bcol[e0 + n - 1];
                    bval[e0 + i] = bval[e0 + n - 1];
                    break;
                }
            bcnt[row] = n - 1;
        }
    __syncthreads();
    if (cj >= 0)
        for (long long e = cptr[cj] + threadIdx.x; e < cptr[cj + 1]; e += blockDim.x) {
            const int row = ridx[e];
            const int n = bcnt[row];
            bcol[rptr[row] + n] = cj;
            bval[rptr[row] + n] = cval[e];
            bcnt[row] = n + 1;
        }
    // live-entry lists of the columns: the entering slack's row leaves R (the leaving slack's row
    // joined in k_sp_btran).  The last entry of a list moves into the hole.
    if (lcnt && cj < 0) {
        const int re = -1 - cj;
        for (long long e = rptr[re] + threadIdx.x; e < rptr[re + 1]; e += blockDim.x) {
            const int col = cidx[e];
            const long long base = cptr[col];
            const int n = lcnt[col];
            for (int i = 0; i < n; ++i)
                if (lent[base + i].row == re) {
                    lent[base + i] = lent[base + n - 1];
                    lcnt[col] = n - 1;
                    break;
                }
        }
    }
}


// ---------------------------------------------------------------------------------
// k_sp_update: the data moves k_sp_pivot booked (no cell is both a source and a target of the
// same pivot: the last row / column are only read), the eta of this pivot in the NEW numbering,
// pivot() x4 (src/simplex.rs:262-265,:410-421) and the first-pivot candidates of the next
// iteration (:423-437).  grid = SP_NB_UPD workgroups of 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sp_update(
    const DzgCtl *ctl, int only_partials, double *x, double *xbar, double *z, double *zbar,
    const double *__restrict__ dx, const double *__restrict__ dz, int m, int q,
    double *fpx_r, int *fpx_k, double *fpx_h, double *fpz_r, int *fpz_k, double *fpz_h,
    const double *__restrict__ v, double *__restrict__ U, long long ldu, double *__restrict__ W,
    long long ldw, double *__restrict__ X, long long ldb, const int *__restrict__ drow,
    const int *__restrict__ spos)
{
    // this thread's first element of each vector leaves beside the control block (nothing here is
    // written by another thread of this launch): one trip instead of two
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    double x_f = 0.0, xb_f = 0.0, dx_f = 0.0, z_f = 0.0, zb_f = 0.0, dz_f = 0.0;
    if (gid < m) {
        x_f = x[gid];
        xb_f = xbar[gid];
        if (!only_partials) dx_f = dx[gid];
    }
    if (gid < q) {
        z_f = z[gid];
        zb_f = zbar[gid];
        if (!only_partials) dz_f = dz[gid];
    }
    const DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    const int p = c.leave_pos, r = c.enter_pos;
    if (!only_partials) {
        const int ko = c.sp_k, kn = c.ncompact, last = ko - 1;
        const int nold = c.neta - 1; // pending etas before this pivot
        const int mrow = c.sp_mrow, mcol = c.sp_mcol;
        if (mrow >= 0) { // row `last` moves into the hole (its entry in column mcol: see below)
            for (int cc = gid; cc < ko; cc += stride)
                if (cc != mcol && cc != last)
                    X[(long long)mrow * ldb + cc] = X[(long long)last * ldb + cc];
            for (int t = gid; t < nold; t += stride)
                U[(long long)t * ldu + mrow] = U[(long long)t * ldu + last];
        }
        if (mcol >= 0) { // column `last` moves into the hole; the moved row takes it from (last, last)
            for (int b = gid; b < last; b += stride) {
                const int src = (b == mrow) ? last : b;
                X[(long long)b * ldb + mcol] = X[(long long)src * ldb + last];
            }
            for (int t = gid; t < nold; t += stride)
                W[(long long)t * ldw + mcol] = W[(long long)t * ldw + last];
        } else if (mrow >= 0 && c.sp_app < 0 && c.sp_zcol < 0 && kn < ko) {
            // the deleted column WAS the last one: the moved row's entry there is dropped
        }
        if (c.sp_zcol >= 0) { // recycled column slot: the new column is zero before this eta
            for (int b = gid; b < ko; b += stride) X[(long long)b * ldb + c.sp_zcol] = 0.0;
            for (int t = gid; t < nold; t += stride) W[(long long)t * ldw + c.sp_zcol] = 0.0;
        }
        if (c.sp_app >= 0) { // new row = v (the row of B^-1 the slack position had), new column 0
            const int a = c.sp_app;
            for (int b = gid; b < a; b += stride) X[(long long)b * ldb + a] = 0.0;
            for (int cc = gid; cc <= a; cc += stride) X[(long long)a * ldb + cc] = v[drow[cc]];
            for (int t = gid; t < nold; t += stride) {
                U[(long long)t * ldu + a] = 0.0;
                W[(long long)t * ldw + a] = 0.0;
            }
        }
        // eta of this pivot: u = (dx - e_p) / dx_p on the structural positions, w = v_R
        const double rdxp = 1.0 / dx[p];
        double *ut = U + (long long)nold * ldu, *wt = W + (long long)nold * ldw;
        for (int b = gid; b < kn; b += stride) {
            const int i = spos[b];
            const double d = dx[i];
            ut[b] = (i == p ? d - 1.0 : d) * rdxp;
            wt[b] = v[drow[b]];
        }
    }
    const double t = c.t, s = c.s, tbar = c.tbar, sbar = c.sbar;
    const double tau = c.tau;
    DzgCand2 bx = dzg_cand2_none(), bz = dzg_cand2_none();
    for (int i = gid; i < m; i += stride) {
        double xi = i == gid ? x_f : x[i], xb = i == gid ? xb_f : xbar[i];
        if (!only_partials) {
            const double d = i == gid ? dx_f : dx[i];
            const double a = t * d, b = tbar * d;
            xi = (i == p) ? t : xi - a;
            xb = (i == p) ? tbar : xb - b;
            x[i] = xi;
            xbar[i] = xb;
        }
        dzg_first_pivot_entry(bx, xi, xb, i, tau);
    }
    for (int kk = gid; kk < q; kk += stride) {
        double zk = kk == gid ? z_f : z[kk], zb = kk == gid ? zb_f : zbar[kk];
        if (!only_partials) {
            const double d = kk == gid ? dz_f : dz[kk];
            const double a = s * d, b = sbar * d;
            zk = (kk == r) ? s : zk - a;
            zb = (kk == r) ? sbar : zb - b;
            z[kk] = zk;
            zbar[kk] = zb;
        }
        dzg_first_pivot_entry(bz, zk, zb, kk, tau);
    }
    bx = dzg_block_best2(bx);
    bz = dzg_block_best2(bz);
    if (threadIdx.x == 0) {
        fpx_r[blockIdx.x] = bx.r;
        fpx_k[blockIdx.x] = bx.k;
        fpx_h[blockIdx.x] = bx.h;
        fpz_r[blockIdx.x] = bz.r;
        fpz_k[blockIdx.x] = bz.k;
        fpz_h[blockIdx.x] = bz.h;
    }
}

// ---------------------------------------------------------------------------------
// Flush: X[0:k, 0:k] -= Ub^T[:, 0:neta] * Wc.  Same tiling as k_fast_flush_mfma (one wave owns a
// 16 x 64 strip, v_mfma_f64_16x16x4_f64 along the eta index); here both operands are compact.
// ---------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

#define SPF_LD 80 // LDS row stride of the W tile in doubles: rows t and t + 1 sixteen banks apart
__global__ __launch_bounds__(256) void k_sp_flush_mfma(const DzgCtl *ctl, double *__restrict__ X,
                                                       long long ldb, const double *__restrict__ U,
                                                       long long ldu, const double *__restrict__ W,
                                                       long long ldw)
{
    // (a flush folds a FULL eta file, like k_fast_flush_mfma: one enqueued behind an iteration that
    // did not pivot is a no-op)
    __shared__ double s_w[R_ * SPF_LD];
    const int neta = ctl->neta, k = ctl->ncompact;
    if (neta < R_ || k <= 0 || ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64;
    const int i0 = (blockIdx.y * 4 + wave) * 16;
    if (c0 >= k || blockIdx.y * 64 >= k) return; // (workgroup-uniform: the barrier below is safe)
    const int li = lane & 15, lk = lane >> 4;
    // Everything the workgroup needs leaves in ONE trip: the 64 x 64 tile of W the four waves share
    // (through LDS), this wave's 16 x 64 strip of U^T and its 16 x 64 strip of X.  (Step by step --
    // a load, a wait, four MFMAs, sixteen times over -- a flush took 131 us at k = 1 175, 9 us
    // this way.)  Rows / columns beyond k are clamped to k - 1: a row of the product depends
    // on its own row of U only, a column on its own column of W, and neither is stored.
    double wreg[16];
    {
        const int c = min(c0 + lane, k - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) wreg[i] = W[(long long)(wave + 4 * i) * ldw + c];
    }
    double a[16];
    {
        const int arow = min(i0 + li, k - 1);
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) a[s4] = -U[(long long)(4 * s4 + lk) * ldu + arow];
    }
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = min(i0 + lk + 4 * g, k - 1), col = min(c0 + 16 * j + li, k - 1);
            acc[j][g] = X[(long long)row * ldb + col];
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) s_w[(wave + 4 * i) * SPF_LD + lane] = wreg[i];
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
        const int t = 4 * s4 + lk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], s_w[t * SPF_LD + 16 * j + li], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            if (row < k && col < k) X[(long long)row * ldb + col] = acc[j][g];
        }
}

// kcap: the block width the flush's grid covered (the host's bound on k for the batch, k_hint).
// A wider block would have been flushed in part only: loud, not silent.
__global__ void k_sp_flush_done(DzgCtl *ctl, int kcap)
{
    if (ctl->status == DZG_RUNNING && ctl->neta >= R_) {
        if (ctl->ncompact > kcap) ctl->status = DZG_PANIC;
        ctl->neta = 0;
    }
}

// ---------------------------------------------------------------------------------
// Books of the starting basis (after k_fast_init has listed the dense columns): which basis
// positions hold structurals (rows of X, in position order), where each slack is basic.
// One workgroup; runs once per solve and after every refactorisation.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sp_init(DzgCtl *ctl, int m, int ns,
                                                 const int *__restrict__ basis,
                                                 const int *__restrict__ var_col, int *sslot,
                                                 int *spos, int *bslot, int *rowpos, int *acol_code)
{
    for (int j = threadIdx.x; j < ns; j += blockDim.x) bslot[j] = -1;
    for (int r = threadIdx.x; r < m; r += blockDim.x) rowpos[r] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int b = 0;
        for (int p = 0; p < m; ++p) {
            const int code = var_col[basis[p]];
            if (code >= 0) {
                sslot[p] = b;
                spos[b] = p;
                bslot[code] = b;
                ++b;
            } else {
                sslot[p] = -1;
                rowpos[-1 - code] = p;
            }
        }
        if (acol_code) *acol_code = (int)0x80000000;
        ctl->rl_listed = -1; // (k_sp_rlists, launched next, lists the rows of R and nothing else)
    }
}

// basic-entry lists from scratch (creation with a non-slack basis, after a refactorisation):
// row r keeps, in CSR order, its entries whose column is basic.  grid over rows.
__global__ __launch_bounds__(256) void k_sp_lists(int m, const long long *__restrict__ rptr,
                                                  const int *__restrict__ cidx,
                                                  const double *__restrict__ rval,
                                                  const int *__restrict__ bslot, int *bcnt,
                                                  int *bcol, double *bval)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < m; r += gridDim.x * blockDim.x) {
        const long long e0 = rptr[r];
        int n = 0;
        for (long long e = e0; e < rptr[r + 1]; ++e)
            if (bslot[cidx[e]] >= 0) {
                bcol[e0 + n] = cidx[e];
                bval[e0 + n] = rval[e];
                ++n;
            }
        bcnt[r] = n;
    }
}

// live-entry lists from scratch (creation, after a refactorisation): column j keeps, in CSC
// order, its entries in rows of R (dslot >= 0).  grid over columns.
__global__ __launch_bounds__(256) void k_sp_rlists(int ns, const long long *__restrict__ cptr,
                                                   const int *__restrict__ ridx,
                                                   const double *__restrict__ cval,
                                                   const int *__restrict__ dslot, int *lcnt,
                                                   DzgLiveEntry *lent)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ns; j += gridDim.x * blockDim.x) {
        const long long e0 = cptr[j];
        int n = 0;
        for (long long e = e0; e < cptr[j + 1]; ++e)
            if (dslot[ridx[e]] >= 0) {
                DzgLiveEntry en;
                en.row = ridx[e];
                en.pad_ = 0;
                en.val = cval[e];
                lent[e0 + n] = en;
                ++n;
            }
        lcnt[j] = n;
    }
}

// after a refactorisation: Xinv (row b = b-th structural basic in position order, column a =
// compact column a) becomes X.  grid (ceil(k / 256), k)
__global__ __launch_bounds__(256) void k_sp_ref_copy(int k, const double *__restrict__ Xinv,
                                                     long long ldx, double *__restrict__ X,
                                                     long long ldb)
{
    const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < k) X[(long long)b * ldb + a] = Xinv[(long long)b * ldx + a];
}


// =================================================================================
// The same iteration in FOUR launches instead of eight (round 4):
//
//      k_sp_pre    status();  a primal step: FTRAN on X's rows | barrier | the basic slacks' rows +
//                  ratio candidates | barrier | ratio decision;   then BTRAN's row (+ live lists)
//      pricing     k_price_csc_rl, unchanged
//      k_sp_mid    a dual step: ratio decision, FTRAN on X's rows | barrier | the basic slacks'
//                  rows | barrier;   then the pivot's books (workgroup 0)
//      k_sp_update unchanged (the books' index maps reach it across the kernel boundary)
//
// The phases are the bodies of the kernels above, glued with the device-wide barrier of the dense
// chain (chain_barrier.h: fence-free, fails consistently, every wave reaches its exit).  What
// crosses a barrier inside a launch -- the dense copy of the entering column (written by workgroup
// 0), dx on the structural rows of X (`dxs`, gathered by the slack rows), the ratio candidates, and
// in a dual step dx at the leaving position for the books -- is written with agent-scope (sc1)
// stores, drained by every storing wave before the workgroup arrives, and read with sc1 loads.
// Every workgroup takes every decision itself from the same partial results, so no decision
// crosses a barrier.  Nothing of an iteration's STATE is written before the last barrier of a launch
// (the live-list append of BTRAN and the books come after it), so a failed barrier leaves the state
// of the last completed pivot and the host carries on with the eight launches (engine.hip).
// Same arithmetic in the same order as the eight-launch form: the same pivots, bit for bit.
// grid = sp_grid(m) workgroups of 256, all resident (checked at creation).
// =================================================================================
struct SpStage {
    int *s_slot;
    double *s_val;
    int *s_wcnt;
};

// the entering column's entries in rows of R, 256 at a time, compacted in their order (see
// k_sp_ftran_s); returns the count
__device__ __forceinline__ int sp_stage_chunk(const DzgDev &d, const SpStage &st, int code,
                                              long long base, long long e1)
{
    __syncthreads(); // the previous chunk has been consumed
    const long long e = base + threadIdx.x;
    int slot = -1;
    double val = 0.0;
    if (e < e1) {
        const int r = code >= 0 ? d.ridx[e] : -1 - code;
        slot = d.dslot[r];
        val = code >= 0 ? d.cval[e] : 1.0;
    }
    const unsigned long long mask = __ballot(slot >= 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) st.s_wcnt[wave] = __popcll(mask);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += st.s_wcnt[w];
    if (slot >= 0) {
        const int at = off + __popcll(mask & ((1ull << lane) - 1ull));
        st.s_slot[at] = slot;
        st.s_val[at] = val;
    }
    const int total = st.s_wcnt[0] + st.s_wcnt[1] + st.s_wcnt[2] + st.s_wcnt[3];
    __syncthreads();
    return total;
}

// FTRAN of the entering column `code` (both halves around one barrier).  PRIMAL: also the ratio
// candidates of this workgroup's rows in `best`.  dx at every position is published with sc1 stores.
// Returns false when the barrier failed.
template <int KIND>
__device__ __forceinline__ bool sp_ftran_fused(const DzgDev &d, DzgCtl *ctl, unsigned long long *bar,
                                               unsigned long long &gen, int code, int k, int neta,
                                               double mu, double tau, const SpStage &st,
                                               double *s_beta, DzgCand2 &best)
{
    const long long e0 = code >= 0 ? d.cptr[code] : 0, e1 = code >= 0 ? d.cptr[code + 1] : 1;
    if (blockIdx.x == 0) { // the dense copy of the entering column, for the basic slacks' rows
        const int prev = *d.acol_code; // INT_MIN: nothing scattered yet
        if (prev != (int)0x80000000) {
            if (prev < 0) {
                if (threadIdx.x == 0) st_sc1(d.acol + (-1 - prev), 0.0);
            } else {
                for (long long e = d.cptr[prev] + threadIdx.x; e < d.cptr[prev + 1]; e += blockDim.x)
                    st_sc1(d.acol + d.ridx[e], 0.0);
            }
        }
        __syncthreads(); // the two columns may share rows
        if (code < 0) {
            if (threadIdx.x == 0) st_sc1(d.acol + (-1 - code), 1.0);
        } else {
            for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) st_sc1(d.acol + d.ridx[e], d.cval[e]);
        }
        if (threadIdx.x == 0) *d.acol_code = code;
    }
    // ---- beta_t = W_t . a_R, thread t (k_sp_ftran_s)
    double bacc = 0.0;
    int cnt = 0;
    for (long long base = e0; base < e1; base += 256) {
        cnt = sp_stage_chunk(d, st, code, base, e1);
        if ((int)threadIdx.x < neta) {
            const double *wt = d.W + (long long)threadIdx.x * d.ldw;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double w0 = wt[st.s_slot[i]], w1 = wt[st.s_slot[i + 1]], w2 = wt[st.s_slot[i + 2]],
                             w3 = wt[st.s_slot[i + 3]];
                bacc = fma(w0, st.s_val[i], bacc);
                bacc = fma(w1, st.s_val[i + 1], bacc);
                bacc = fma(w2, st.s_val[i + 2], bacc);
                bacc = fma(w3, st.s_val[i + 3], bacc);
            }
            for (; i < cnt; ++i) bacc = fma(wt[st.s_slot[i]], st.s_val[i], bacc);
        }
    }
    if (threadIdx.x < R_) s_beta[threadIdx.x] = (int)threadIdx.x < neta ? bacc : 0.0;
    __syncthreads();
    // ---- dx on the rows of X
    const bool one_chunk = e1 - e0 <= 256;
    for (int b0 = blockIdx.x * blockDim.x; b0 < k; b0 += gridDim.x * blockDim.x) { // block-uniform
        const int b = b0 + threadIdx.x;
        const double *row = d.binv + (long long)(b < k ? b : 0) * d.ldb;
        const int i = b < k ? d.spos[b] : 0;
        double xi = 0.0, xbi = 0.0;
        if (KIND == DZG_STEP_PRIMAL && b < k) {
            xi = d.x[i];
            xbi = d.xbar[i];
        }
        double acc = 0.0;
        for (long long base = e0; base < e1; base += 256) {
            if (!one_chunk) cnt = sp_stage_chunk(d, st, code, base, e1);
            else __syncthreads();
            if (b < k) {
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const double r0 = row[st.s_slot[j]], r1 = row[st.s_slot[j + 1]], r2 = row[st.s_slot[j + 2]],
                                 r3 = row[st.s_slot[j + 3]];
                    acc = fma(r0, st.s_val[j], acc);
                    acc = fma(r1, st.s_val[j + 1], acc);
                    acc = fma(r2, st.s_val[j + 2], acc);
                    acc = fma(r3, st.s_val[j + 3], acc);
                }
                for (; j < cnt; ++j) acc = fma(row[st.s_slot[j]], st.s_val[j], acc);
            }
        }
        if (b < k) {
            int t = 0;
            for (; t + 8 <= neta; t += 8) {
                double u[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) u[g] = d.U[(long long)(t + g) * d.ldw + b];
#pragma unroll
                for (int g = 0; g < 8; ++g) acc = fma(-u[g], s_beta[t + g], acc);
            }
            for (; t < neta; ++t) acc = fma(-d.U[(long long)t * d.ldw + b], s_beta[t], acc);
            st_sc1(d.dxs + b, acc);
            st_sc1(d.dx + i, acc);
            if (KIND == DZG_STEP_PRIMAL) {
                const double scaled = mu * xbi;
                const double den = xi + scaled;
                DzgCand2 cnd;
                cnd.r = dzg_div(acc, den);
                cnd.k = i;
                cnd.h = -__builtin_inf();
                if (cnd.r > 0.0) best = dzg_better2(best, cnd);
                if (dzg_noise_zero(den, xi, scaled, tau)) best.h = __builtin_inf();
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0); // every storing wave drains its sc1 stores before the workgroup arrives
    if (!chain_barrier(ctl, bar, gen)) return false;
    // ---- dx on the positions of the basic slacks (k_sp_ftran_l): row lists in list order
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < d.m; r += gridDim.x * blockDim.x) {
        const int p = d.rowpos[r];
        if (p < 0) continue;
        double acc = ld_sc1(d.acol + r);
        const long long l0 = d.rptr[r];
        const int n = d.bcnt[r];
        double xi = 0.0, xbi = 0.0;
        if (KIND == DZG_STEP_PRIMAL) {
            xi = d.x[p];
            xbi = d.xbar[p];
        }
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const int c0 = d.bcol[l0 + i], c1 = d.bcol[l0 + i + 1], c2 = d.bcol[l0 + i + 2], c3 = d.bcol[l0 + i + 3];
            const double v0 = d.bval[l0 + i], v1 = d.bval[l0 + i + 1], v2 = d.bval[l0 + i + 2], v3 = d.bval[l0 + i + 3];
            const int s0 = d.bslot[c0], s1 = d.bslot[c1], s2 = d.bslot[c2], s3 = d.bslot[c3];
            const double d0 = ld_sc1(d.dxs + s0), d1 = ld_sc1(d.dxs + s1), d2 = ld_sc1(d.dxs + s2),
                         d3 = ld_sc1(d.dxs + s3);
            acc = fma(-v0, d0, acc);
            acc = fma(-v1, d1, acc);
            acc = fma(-v2, d2, acc);
            acc = fma(-v3, d3, acc);
        }
        for (; i < n; ++i) acc = fma(-d.bval[l0 + i], ld_sc1(d.dxs + d.bslot[d.bcol[l0 + i]]), acc);
        st_sc1(d.dx + p, acc);
        if (KIND == DZG_STEP_PRIMAL) {
            const double scaled = mu * xbi;
            const double den = xi + scaled;
            DzgCand2 cnd;
            cnd.r = dzg_div(acc, den);
            cnd.k = p;
            cnd.h = -__builtin_inf();
            if (cnd.r > 0.0) best = dzg_better2(best, cnd);
            if (dzg_noise_zero(den, xi, scaled, tau)) best.h = __builtin_inf();
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void k_sp_pre(const DzgDev d, unsigned long long *bar)
{
    __shared__ int s_slot[256];
    __shared__ double s_val[256];
    __shared__ double s_beta[R_];
    __shared__ int s_wcnt[4];
    __shared__ int s_b[SP_LCAP];
    __shared__ double s_coef[SP_LCAP];
    __shared__ double s_gamma[R_];
    __shared__ int s_cnt;
    DzgCtl *ctl = d.ctl;
    const DzgCand2 cj = reduce_partials(d.fpz_r, d.fpz_k, d.fpz_h, SP_NB_UPD);
    const DzgCand2 ci = reduce_partials(d.fpx_r, d.fpx_k, d.fpx_h, SP_NB_UPD);
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    unsigned long long gen = c.bar_gen;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int m = d.m, tid = threadIdx.x;
    int kind;
    double mu;
    if (!fast_status(ctl, c, lead, cj, ci, d.eps, m, false, kind, &mu)) return;
    const int k = c.ncompact, neta = c.neta;
    int p;
    if (kind == DZG_STEP_PRIMAL) {
        const int code = d.nbcode[cj.k];
        if (lead) ctl->enter_code = code;
        const SpStage st{s_slot, s_val, s_wcnt};
        DzgCand2 best = dzg_cand2_none();
        if (!sp_ftran_fused<DZG_STEP_PRIMAL>(d, ctl, bar, gen, code, k, neta, mu, c.tau, st, s_beta, best)) return;
        best = dzg_block_best2(best);
        if (tid == 0) {
            st_sc1(d.rx_r + blockIdx.x, best.r);
            st_sc1(d.rx_k + blockIdx.x, best.k);
            st_sc1(d.rx_h + blockIdx.x, best.h);
        }
        __builtin_amdgcn_s_waitcnt(0); // (dx of the slack rows: later launches read it plainly)
        if (!chain_barrier(ctl, bar, gen)) return;
        DzgCand2 w = dzg_cand2_none();
        for (int i = tid; i < (int)gridDim.x; i += blockDim.x) {
            DzgCand2 o;
            o.r = ld_sc1(d.rx_r + i);
            o.k = ld_sc1(d.rx_k + i);
            o.h = ld_sc1(d.rx_h + i);
            w = dzg_better2(w, o);
        }
        const DzgCand2 cw = dzg_block_best2(w);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_UNBOUNDED)) { // src/simplex.rs:313
            if (lead) ctl->bar_gen = gen;
            return;
        }
        p = cw.k;
        if (lead) ctl->leave_pos = p;
    } else {
        p = ci.k;
    }
    if (lead && gen != c.bar_gen) ctl->bar_gen = gen;
    // ---- BTRAN: v = row p of B^-1 in row coordinates (k_sp_btran's body)
    const int bp = d.sslot[p];
    const int rl = bp >= 0 ? -1 : -1 - d.bcode[p];
    const int stride = gridDim.x * blockDim.x;
    const int gid = blockIdx.x * blockDim.x + tid;
    if (d.lcnt && blockIdx.x == 0 && c.rl_listed != rl) { // live-entry lists: idempotent append
        const int pend = c.rl_listed;
        if (pend >= 0) {
            for (long long e = d.rptr[pend] + tid; e < d.rptr[pend + 1]; e += blockDim.x) {
                const int col = d.cidx[e];
                const long long base = d.cptr[col];
                const int n = d.lcnt[col];
                for (int i = 0; i < n; ++i)
                    if (d.lent[base + i].row == pend) {
                        d.lent[base + i] = d.lent[base + n - 1];
                        d.lcnt[col] = n - 1;
                        break;
                    }
            }
            __syncthreads();
        }
        if (rl >= 0)
            for (long long e = d.rptr[rl] + tid; e < d.rptr[rl + 1]; e += blockDim.x) {
                const int col = d.cidx[e];
                const long long at = d.cptr[col] + d.lcnt[col];
                DzgLiveEntry en;
                en.row = rl;
                en.pad_ = 0;
                en.val = d.rval[e];
                d.lent[at] = en;
                d.lcnt[col] += 1;
            }
        if (tid == 0) ctl->rl_listed = rl;
    }
    for (int r = gid; r < m; r += stride)
        if (d.dslot[r] < 0) d.v[r] = (r == rl) ? 1.0 : 0.0;
    if (k == 0) return;
    const long long e0 = bp >= 0 ? 0 : d.rptr[rl], e1 = bp >= 0 ? 1 : e0 + d.bcnt[rl];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (tid < R_) s_gamma[tid] = 0.0;
    for (long long base = e0; base < e1; base += SP_LCAP) {
        __syncthreads();
        int total = 0;
        if (bp >= 0) {
            if (tid == 0) {
                s_b[0] = bp;
                s_coef[0] = 1.0;
            }
            total = 1;
        } else {
            const long long lim = (e1 - base) < SP_LCAP ? (e1 - base) : SP_LCAP;
            for (long long i = tid; i < lim; i += blockDim.x) {
                s_b[i] = d.bslot[d.bcol[base + i]];
                s_coef[i] = -d.bval[base + i];
            }
            total = (int)lim;
        }
        if (tid == 0) s_cnt = total;
        __syncthreads();
        const int cnt = s_cnt;
        if (tid < neta) {
            double g = s_gamma[tid];
            const double *ut = d.U + (long long)tid * d.ldw;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double u0 = ut[s_b[i]], u1 = ut[s_b[i + 1]], u2 = ut[s_b[i + 2]], u3 = ut[s_b[i + 3]];
                g = fma(s_coef[i], u0, g);
                g = fma(s_coef[i + 1], u1, g);
                g = fma(s_coef[i + 2], u2, g);
                g = fma(s_coef[i + 3], u3, g);
            }
            for (; i < cnt; ++i) g = fma(s_coef[i], ut[s_b[i]], g);
            s_gamma[tid] = g;
        }
        int slot = 0;
        for (int cc = gid; cc < k && slot < 4; cc += stride, ++slot) {
            double a = acc[slot];
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const double x0 = d.binv[(long long)s_b[i] * d.ldb + cc], x1 = d.binv[(long long)s_b[i + 1] * d.ldb + cc],
                             x2 = d.binv[(long long)s_b[i + 2] * d.ldb + cc], x3 = d.binv[(long long)s_b[i + 3] * d.ldb + cc];
                a = fma(s_coef[i], x0, a);
                a = fma(s_coef[i + 1], x1, a);
                a = fma(s_coef[i + 2], x2, a);
                a = fma(s_coef[i + 3], x3, a);
            }
            for (; i < cnt; ++i) a = fma(s_coef[i], d.binv[(long long)s_b[i] * d.ldb + cc], a);
            acc[slot] = a;
        }
    }
    __syncthreads();
    int slot = 0;
    for (int cc = gid; cc < k; cc += stride, ++slot) {
        double a = slot < 4 ? acc[slot] : 0.0;
        const int vr = d.drow[cc];
        int t = 0;
        for (; t + 8 <= neta; t += 8) {
            double w[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) w[g] = d.W[(long long)(t + g) * d.ldw + cc];
#pragma unroll
            for (int g = 0; g < 8; ++g) a = fma(-s_gamma[t + g], w[g], a);
        }
        for (; t < neta; ++t) a = fma(-s_gamma[t], d.W[(long long)t * d.ldw + cc], a);
        d.v[vr] = a;
    }
}

__global__ __launch_bounds__(256) void k_sp_mid(const DzgDev d, unsigned long long *bar, int nrz)
{
    __shared__ int s_slot[256];
    __shared__ double s_val[256];
    __shared__ double s_beta[R_];
    __shared__ int s_wcnt[4];
    __shared__ int s_ok, s_ci, s_cj;
    DzgCtl *ctl = d.ctl;
    const DzgCand2 cw = reduce_partials(d.rz_r, d.rz_k, d.rz_h, nrz); // (a primal step ignores them)
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    unsigned long long gen = c.bar_gen;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const bool dual = c.kind == DZG_STEP_DUAL;
    if (dual) {
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) return; // src/simplex.rs:325
        const int epos = cw.k;
        const int code = d.nbcode[epos];
        if (lead) {
            ctl->enter_pos = epos;
            ctl->enter_code = code;
        }
        c.enter_pos = epos;
        c.enter_code = code;
        const SpStage st{s_slot, s_val, s_wcnt};
        DzgCand2 unused = dzg_cand2_none();
        if (!sp_ftran_fused<DZG_STEP_DUAL>(d, ctl, bar, gen, code, c.ncompact, c.neta, c.mu, c.tau, st, s_beta, unused))
            return;
        __builtin_amdgcn_s_waitcnt(0);
        if (!chain_barrier(ctl, bar, gen)) return; // dx is complete: the books read it at the leaving position
        if (lead) ctl->bar_gen = gen;
    }
    if (blockIdx.x != 0) return;
    // ---- the pivot's books (k_sp_pivot's body, workgroup 0)
    if (threadIdx.x == 0) {
        const int p = c.leave_pos, r = c.enter_pos;
        const int vi = d.basis[p], vj = d.nonbasis[r];
        s_ci = d.bcode[p];
        s_cj = d.nbcode[r];
        const double dxp = dual ? ld_sc1(d.dx + p) : d.dx[p];
        s_ok = sp_pivot_books(ctl, c, d.m, d.q, p, r, c.neta, vi, vj, s_ci, s_cj, d.x, d.xbar, d.z, d.zbar, d.dx, d.dz,
                              d.basis, d.nonbasis, d.var_col, d.drow, d.dslot, d.sslot, d.spos, d.bslot, d.rowpos,
                              d.plist, d.pslot, d.cptr, d.log_kind, d.log_enter, d.log_leave, d.log_mu,
                              d.log_margin, d.log_cap, d.bcode, d.nbcode, d.pcode, d.lcnt != nullptr, dxp);
    }
    __syncthreads();
    if (!s_ok) return;
    const int ci = s_ci, cj = s_cj;
    if (ci >= 0)
        for (long long e = d.cptr[ci] + threadIdx.x; e < d.cptr[ci + 1]; e += blockDim.x) {
            const int row = d.ridx[e];
            const long long l0 = d.rptr[row];
            const int n = d.bcnt[row];
            for (int i = 0; i < n; ++i)
                if (d.bcol[l0 + i] == ci) {
                    d.bcol[l0 + i] = d.bcol[l0 + n - 1];
                    d.bval[l0 + i] = d.bval[l0 + n - 1];
                    break;
                }
            d.bcnt[row] = n - 1;
        }
    __syncthreads();
    if (cj >= 0)
        for (long long e = d.cptr[cj] + threadIdx.x; e < d.cptr[cj + 1]; e += blockDim.x) {
            const int row = d.ridx[e];
            const int n = d.bcnt[row];
            d.bcol[d.rptr[row] + n] = cj;
            d.bval[d.rptr[row] + n] = d.cval[e];
            d.bcnt[row] = n + 1;
        }
    if (d.lcnt && cj < 0) {
        const int re = -1 - cj;
        for (long long e = d.rptr[re] + threadIdx.x; e < d.rptr[re + 1]; e += blockDim.x) {
            const int col = d.cidx[e];
            const long long base = d.cptr[col];
            const int n = d.lcnt[col];
            for (int i = 0; i < n; ++i)
                if (d.lent[base + i].row == re) {
                    d.lent[base + i] = d.lent[base + n - 1];
                    d.lcnt[col] = n - 1;
                    break;
                }
        }
    }
}

// workgroups of the fused kernels the runtime places on one CU (0: none fits)
int dzg_sp_fused_resident_per_cu(void)
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_sp_pre, 256, 0) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_sp_mid, 256, 0) != hipSuccess) return 0;
    return a < b ? a : b;
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
static int sp_grid(int n) { int g = (n + 255) / 256; return g < 1 ? 1 : (g > SP_NB ? SP_NB : g); }

// first != 0: creation (acol is all zero, nothing scattered); 0: after a refactorisation, when
// acol still holds the last entering column and its code must be remembered
void dzg_launch_sp_init(const DzgDev &d, int first, hipStream_t st)
{
    hipLaunchKernelGGL(k_sp_init, dim3(1), dim3(256), 0, st, d.ctl, d.m, d.ns, d.basis, d.var_col,
                       d.sslot, d.spos, d.bslot, d.rowpos, first ? d.acol_code : (int *)nullptr);
    hipLaunchKernelGGL(k_sp_lists, dim3(sp_grid(d.m)), dim3(256), 0, st, d.m, d.rptr, d.cidx, d.rval,
                       d.bslot, d.bcnt, d.bcol, d.bval);
    if (d.lcnt)
        hipLaunchKernelGGL(k_sp_rlists, dim3(sp_grid(d.ns)), dim3(256), 0, st, d.ns, d.cptr, d.ridx,
                           d.cval, d.dslot, d.lcnt, d.lent);
}

// need_kind = PRIMAL: the head of the iteration (status + primal FTRAN); DUAL: after pricing
// (ratio test with `nrz` pricing partials + dual FTRAN).  Two launches each: structural
// positions (with the head), then the positions of the basic slacks.
void dzg_launch_sp_ftran(const DzgDev &d, int need_kind, int nrz, hipStream_t st)
{
    const int gs = sp_grid(d.m), gl = sp_grid(d.m);
#define SP_FS_ARGS d.ctl, d.m, d.cptr, d.ridx, d.cval, d.nbcode, d.fpx_r, d.fpx_k, d.fpx_h,              \
                   d.fpz_r, d.fpz_k, d.fpz_h, d.rz_r, d.rz_k, d.rz_h, nrz, d.binv, d.ldb, d.U, d.ldw, d.W,  \
                   d.ldw, d.dslot, d.spos, d.x, d.xbar, d.dxs, d.dx, d.rx_r, d.rx_k, d.rx_h, d.acol,        \
                   d.acol_code, d.eps
    if (need_kind == DZG_STEP_PRIMAL)
        hipLaunchKernelGGL((k_sp_ftran_s<DZG_STEP_PRIMAL>), dim3(gs), dim3(256), 0, st, SP_FS_ARGS);
    else
        hipLaunchKernelGGL((k_sp_ftran_s<DZG_STEP_DUAL>), dim3(gs), dim3(256), 0, st, SP_FS_ARGS);
#undef SP_FS_ARGS
    hipLaunchKernelGGL(k_sp_ftran_l, dim3(gl), dim3(256), 0, st, d.ctl, need_kind, d.m, d.rptr,
                       d.bcnt, d.bcol, d.bval, d.bslot, d.rowpos, d.acol, d.dxs, d.x, d.xbar, d.dx,
                       d.rx_r, d.rx_k, d.rx_h, gs);
}

void dzg_launch_sp_btran(const DzgDev &d, hipStream_t st)
{
    const int nparts = 2 * sp_grid(d.m);
    hipLaunchKernelGGL(k_sp_btran, dim3(sp_grid(d.m)), dim3(256), 0, st, d.ctl, d.m, nparts, d.rptr,
                       d.bcnt, d.bcol, d.bval, d.bslot, d.sslot, d.bcode, d.binv, d.ldb, d.U,
                       d.ldw, d.W, d.ldw, d.drow, d.dslot, d.rx_r, d.rx_k, d.rx_h, d.v, d.cptr, d.cidx,
                       d.rval, d.lcnt, d.lent);
}

void dzg_launch_sp_pivot(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_sp_pivot, dim3(1), dim3(256), 0, st, d.ctl, d.m, d.q, d.x, d.xbar, d.z,
                       d.zbar, d.dx, d.dz, d.basis, d.nonbasis, d.var_col, d.drow, d.dslot, d.sslot,
                       d.spos, d.bslot, d.rowpos, d.plist, d.pslot, d.cptr, d.ridx, d.cval, d.rptr, d.bcnt,
                       d.bcol, d.bval, d.log_kind, d.log_enter, d.log_leave, d.log_mu, d.log_margin,
                       d.log_cap, d.bcode, d.nbcode, d.pcode, d.cidx, d.lcnt, d.lent);
}

void dzg_launch_sp_update(const DzgDev &d, int only_partials, hipStream_t st)
{
    hipLaunchKernelGGL(k_sp_update, dim3(SP_NB_UPD), dim3(256), 0, st, d.ctl, only_partials, d.x,
                       d.xbar, d.z, d.zbar, d.dx, d.dz, d.m, d.q, d.fpx_r, d.fpx_k, d.fpx_h, d.fpz_r,
                       d.fpz_k, d.fpz_h, d.v, d.U, d.ldw, d.W, d.ldw, d.binv, d.ldb, d.drow, d.spos);
}

void dzg_launch_sp_flush(const DzgDev &d, hipStream_t st)
{
    // k <= k_hint <= m (the host's bound for the batch in flight); the kernel masks by the device's k
    const int kmax = d.k_hint > 0 && d.k_hint < d.m ? d.k_hint : d.m;
    hipLaunchKernelGGL(k_sp_flush_mfma, dim3((kmax + 63) / 64, (kmax + 63) / 64), dim3(256), 0, st,
                       d.ctl, d.binv, d.ldb, d.U, d.ldw, d.W, d.ldw);
    hipLaunchKernelGGL(k_sp_flush_done, dim3(1), dim3(1), 0, st, d.ctl, ((kmax + 63) / 64) * 64);
}

int dzg_sp_grid(int m) { return sp_grid(m); }

void dzg_launch_sp_pre(const DzgDev &d, unsigned long long *bar, hipStream_t st)
{
    hipLaunchKernelGGL(k_sp_pre, dim3(sp_grid(d.m)), dim3(256), 0, st, d, bar);
}

void dzg_launch_sp_mid(const DzgDev &d, unsigned long long *bar, int nrz, hipStream_t st)
{
    hipLaunchKernelGGL(k_sp_mid, dim3(sp_grid(d.m)), dim3(256), 0, st, d, bar, nrz);
}

void dzg_launch_sp_ref_copy(const DzgDev &d, int k, const double *Xinv, long long ldx, hipStream_t st)
{
    if (k > 0)
        hipLaunchKernelGGL(k_sp_ref_copy, dim3((k + 255) / 256, k), dim3(256), 0, st, k, Xinv, ldx,
                           d.binv, d.ldb);
    dzg_launch_sp_init(d, 0, st); // rows of X are in position order again
}

// k_fast.hip -- FAST numerics: the basis inverse stays resident in HBM.
//
// The reference refactorises B and B^T from scratch in every iteration
// (src/simplex.rs:228,234 -> src/linalg.rs:88-128), (4/3)m^3 flops per pivot, and then runs
// two triangular solves (src/linalg.rs:282-299).  Triangular solves are chains of dependent
// steps -- latency-bound on a GPU -- so this engine keeps an explicit inverse instead and turns
// both solves into bandwidth-bound streaming:
//
//      Binv = Binv0 - U W^T                                   (product form, <= 64 pending etas)
//
//   * Binv0 is stored COMPACT: only the columns that belong to rows whose slack is nonbasic
//     are dense; the column of a row whose slack is basic at position p is the unit vector
//     e_p and is never stored (B = [A_S | E] => B^-1 has the same structure).  The basis
//     starts as the slack identity (src/simplex.rs:190-201), so Binv0 starts EMPTY: no
//     factorisation is needed to start, and FTRAN costs 8*m*k bytes with k = #structural
//     basics instead of 8*m^2.
//   * FTRAN  dx = B^-1 a_j :  one GEMV over the compact Binv0 (one wave per row, coalesced
//     16-B loads), the unit columns contribute a_j[r] at the slack's position, the eta file
//     contributes -U (W^T a_j).
//   * BTRAN  v = B^-T e_p  :  row p of Binv -- a gather from one compact row minus a skinny
//     GEMV over W.  No solve at all.
//   * pivot: append u = (dx - e_p)/dx_p and w = v to the eta file (O(m)); a leaving slack
//     appends a compact column, an entering slack deletes one (swap with the last).
//   * every 64 pivots the eta file is folded into Binv0 by one rank-64 update
//     Binv0 -= U * Wc on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): the only true GEMM
//     on the path.
//
// The selection logic (src/simplex.rs:274-306 status, :423-461 pivot rules) is fused into
// the heads of these kernels: each workgroup reduces the small arrays of per-workgroup
// partial candidates left by the previous kernel (deterministic max-loc, lowest position on
// ties), workgroup 0 publishes the decision in the control block for the next kernel.
#include <cstdlib>

#include "common.h"
#include "fast_decide.h"
#include "fast_rows.h"

// ---------------------------------------------------------------------------------
// k_fast_select_prep<MODE>
//   MODE 0: head of the iteration.  status(): first pivots on both sides from the partials
//           left by k_fast_update, optimality test, primal/dual choice (src/simplex.rs:274-306).
//           If primal: FTRAN preparation for the entering column.
//   MODE 1: dual step after pricing: ratio test on the z side from the pricing kernel's
//           partials (src/simplex.rs:324-325), then FTRAN preparation.
//   FTRAN preparation: workgroup t < neta computes beta_t = W_t . a_j; the last workgroup
//   gathers a_j into compact coordinates (ag[c] = a_j[drow[c]]).
// grid = DZG_RMAX + 1 workgroups of 256.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_fast_select_prep(
    DzgCtl *ctl, int m, const double *__restrict__ A, long long lda, int col0,
    const double *__restrict__ xrecv, long long xstride,
    const int *__restrict__ nbcode,
    const double *__restrict__ fpx_r, const int *__restrict__ fpx_k,
    const double *__restrict__ fpx_h, const double *__restrict__ fpz_r,
    const int *__restrict__ fpz_k, const double *__restrict__ fpz_h,
    const double *__restrict__ rz_r, const int *__restrict__ rz_k,
    const double *__restrict__ rz_h, int nrz, const double *__restrict__ W, long long ldw,
    const int *__restrict__ drow, double *__restrict__ ag, double *__restrict__ beta, double eps,
    int world)
{
    DzgCtl c = *ctl; // one snapshot of the control block (scalar loads)
    if (c.status != DZG_RUNNING) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    int epos;
    int code_known = 0, code_val = -1;
    if (MODE == 0 || MODE == 4) {
        // MODE 0: z-side first pivot from this GPU's partials.  MODE 4 (column sharding): from the
        // merge of every rank's proposal -- all ranks see the same records in the same order and
        // apply the same rule, so they take the same decision without a broadcast.
        DzgCand2 cj;
        int w = -1;
        if (MODE == 4)
            w = shard_merge(xrecv, xstride, world, cj);
        else
            cj = reduce_partials(fpz_r, fpz_k, fpz_h, c.fp_count);
        const DzgCand2 ci = reduce_partials(fpx_r, fpx_k, fpx_h, c.fp_count);
        int kind;
        if (!fast_status(ctl, c, lead, cj, ci, eps, m, MODE == 4, kind)) return;
        if (kind != DZG_STEP_PRIMAL) return;
        epos = cj.k;
        if (MODE == 4) { // the entering column travels in the winner's record
            code_known = 1;
            code_val = (int)xrecv[(long long)w * xstride + 5];
            c.enter_src = w;
            if (lead) ctl->enter_src = w;
        }
    } else if (MODE == 5) {
        // column sharding, second exchange.  Dual step: merge the ratio-test proposals (none =
        // Infeasible, src/simplex.rs:325), take the entering column and z, zbar, dz from the
        // winner's record.  Primal step: only z, zbar, dz of the entering position are needed.
        if (c.kind == DZG_STEP_DUAL) {
            DzgCand2 cw;
            const int w = shard_merge(xrecv, xstride, world, cw);
            const double margin = ratio_margin(cw, c.tau);
            if (tie_gate(ctl, c, lead, margin, false)) return;
            if (w < 0) {
                if (lead) {
                    ctl->status = DZG_INFEASIBLE;
                    tie_book_terminal(ctl, c, margin);
                }
                return;
            }
            const double *rec = xrecv + (long long)w * xstride;
            epos = cw.k;
            code_known = 1;
            code_val = (int)rec[5];
            c.enter_src = w;
            if (lead) {
                ctl->enter_pos = epos;
                ctl->enter_src = w;
                ctl->zr = rec[2];
                ctl->zbar_r = rec[3];
                ctl->dz_r = rec[4];
                ctl->use_record = 1;
            }
        } else {
            if (lead) {
                int w = -1;
                for (int r = 0; r < world && w < 0; ++r)
                    if ((int)xrecv[(long long)r * xstride + 1] == c.enter_pos) w = r;
                if (w < 0) {
                    ctl->status = DZG_PANIC; // no rank owns the entering position: cannot happen
                } else {
                    const double *rec = xrecv + (long long)w * xstride;
                    ctl->zr = rec[2];
                    ctl->zbar_r = rec[3];
                    ctl->dz_r = rec[4];
                    ctl->use_record = 1;
                }
            }
            return; // FTRAN already happened in phase 2
        }
    } else {
        if (c.kind != DZG_STEP_DUAL) return;
        const DzgCand2 cw = reduce_partials(rz_r, rz_k, rz_h, nrz);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) return; // :325
        epos = cw.k;
        if (lead) ctl->enter_pos = epos;
    }
    // ---- FTRAN preparation for the entering variable
    const int code = code_known ? code_val : nbcode[epos];
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->enter_code = code;
    const double *a = dzg_enter_col(&c, code, A, lda, col0, xrecv, xstride);
    const int neta = c.neta, k = c.ncompact;
    const int b = blockIdx.x;
    if (b < R_) {
        if (b >= neta) return;
        const double *wt = W + (long long)b * ldw;
        if (code < 0) {
            if (threadIdx.x == 0) beta[b] = wt[-1 - code];
            return;
        }
        const double acc = fast_beta_dot(wt, a, m);
        if (threadIdx.x == 0) beta[b] = acc;
    } else {
        if (code < 0) {
            const int rr = -1 - code;
            for (int c = threadIdx.x; c < k; c += blockDim.x) ag[c] = (drow[c] == rr) ? 1.0 : 0.0;
        } else {
            for (int c = threadIdx.x; c < k; c += blockDim.x) ag[c] = a[drow[c]];
        }
        // pad to a multiple of 2 so the GEMV can read 16 B at a time
        if (threadIdx.x == 0 && (k & 1)) ag[k] = 0.0;
    }
}

// ---------------------------------------------------------------------------------
// k_fast_gemv: dx = Binv a_j.  LPR lanes cooperate on one row (64 when the compact width k is
// large, 16 when it is small so that a wave covers 4 rows per pass).  A primal step also leaves
// the per-workgroup ratio-test candidates (src/simplex.rs:439-461) for k_fast_btran.
// grid = DZG_NB_GEMV workgroups of 256.
// ---------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ void gemv_rows(const DzgCtl *ctl, int need_kind, int m, int k, int neta,
                                          int code, const double *__restrict__ binv,
                                          long long ldb, const double *__restrict__ ag,
                                          const double *__restrict__ U, long long ldu,
                                          const double *__restrict__ beta,
                                          const double *__restrict__ acolp,
                                          const int *__restrict__ bcode,
                                          const double *__restrict__ x,
                                          const double *__restrict__ xbar,
                                          double *__restrict__ dx, DzgCand2 &best, int want_row,
                                          double *want_dx, bool nt)
{
    constexpr int RPW = 64 / LPR; // rows per wave and pass
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int k2 = (k + 1) & ~1;
    const double mu = ctl->mu, tau = ctl->tau;
    for (int i0 = wave_global * RPW; i0 < m; i0 += nwaves * RPW) {
        const int i = i0 + grp;
        // (one wave per row of a wide inverse: the row streams past the caches, eight steps' loads in
        // flight -- profiles/r04_ftran_row_loads_ab.txt; the sums are the same sums)
        double acc = (LPR == 64 && nt)
                         ? fast_gemv_row<LPR, 2>(i, m, k2, neta, binv, ldb, ag, U, ldu, beta, sub)
                         : fast_gemv_row<LPR, 0>(i, m, k2, neta, binv, ldb, ag, U, ldu, beta, sub);
        if (i < m && sub == 0) {
            acc = fast_gemv_unit(acc, bcode[i], code, acolp);
            dx[i] = acc;
            if (i == want_row) *want_dx = acc; // (LDS: the row whose dx the pivot's books need)
            if (need_kind == DZG_STEP_PRIMAL) {
                const double xi = x[i], scaled = mu * xbar[i];
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
}

template <bool PIVOT>
__global__ __launch_bounds__(256) void k_fast_gemv(
    DzgCtl *ctl, int need_kind, int m, const double *__restrict__ binv, long long ldb,
    const double *__restrict__ ag, const double *__restrict__ U, long long ldu,
    const double *__restrict__ beta, const double *__restrict__ A, long long lda, int col0,
    const double *__restrict__ xrecv, long long xstride, const int *__restrict__ bcode,
    const double *__restrict__ x, const double *__restrict__ xbar, double *__restrict__ dx,
    double *__restrict__ rx_r, int *__restrict__ rx_k, double *__restrict__ rx_h,
    DzgPivotArgs pa, int nt_k)
{
    const DzgCtl c = *ctl; // one snapshot of the control block (scalar loads)
    if (c.status != DZG_RUNNING) return;
    __shared__ double s_dxp;
    if (c.kind != need_kind) {
        // PIVOT (the dual-step launch) in a primal step: dx has been there since the primal
        // launch, the ratio tests are done: workgroup 0 keeps the books of the pivot
        if (PIVOT && blockIdx.x == 0) fast_pivot_books(ctl, c, pa, dx[c.leave_pos]);
        return;
    }
    const int k = c.ncompact, neta = c.neta;
    const int code = c.enter_code;
    const double *acolp = dzg_enter_col(&c, code, A, lda, col0, xrecv, xstride);
    DzgCand2 best = dzg_cand2_none();
    // dual step: the workgroup that computes row p of dx keeps the books right after (no other
    // workgroup's result is needed, so no device-wide synchronisation: a release fence of 512
    // workgroups costs ~30 us here)
    const int p = PIVOT ? c.leave_pos : -1;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int rpw = k > 512 ? 1 : 4;
    const bool owner = PIVOT && ((p / rpw) % nwaves) / (int)(blockDim.x >> 6) == (int)blockIdx.x;
    if (k > 512)
        gemv_rows<64>(&c, need_kind, m, k, neta, code, binv, ldb, ag, U, ldu, beta, acolp, bcode,
                      x, xbar, dx, best, p, &s_dxp, k >= nt_k);
    else
        gemv_rows<16>(&c, need_kind, m, k, neta, code, binv, ldb, ag, U, ldu, beta, acolp, bcode,
                      x, xbar, dx, best, p, &s_dxp, false);
    if (need_kind == DZG_STEP_PRIMAL) {
        best = dzg_block_best2(best);
        if (threadIdx.x == 0) {
            rx_r[blockIdx.x] = best.r;
            rx_k[blockIdx.x] = best.k;
            rx_h[blockIdx.x] = best.h;
        }
    }
    if (owner) { // block-uniform
        __syncthreads();
        fast_pivot_books(ctl, c, pa, s_dxp);
    }
}

// ---------------------------------------------------------------------------------
// k_fast_btran: v = row p of Binv.  A primal step first finishes its ratio test (leaving
// position p = argmax over the GEMV partials; none = Unbounded, src/simplex.rs:313).
// grid = ceil(m / 256) workgroups of 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fast_btran(
    DzgCtl *ctl, int m, const double *__restrict__ binv, long long ldb,
    const int *__restrict__ dslot, const int *__restrict__ bcode,
    const double *__restrict__ U, long long ldu, const double *__restrict__ W, long long ldw,
    const double *__restrict__ rx_r, const int *__restrict__ rx_k,
    const double *__restrict__ rx_h, double *__restrict__ v, double *__restrict__ vc)
{
    __shared__ double s_up[R_];
    DzgCtl c = *ctl; // one snapshot of the control block (scalar loads)
    if (c.status != DZG_RUNNING) return;
    int p;
    if (c.kind == DZG_STEP_PRIMAL) {
        const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
        const DzgCand2 cw = reduce_partials(rx_r, rx_k, rx_h, DZG_NB_GEMV);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_UNBOUNDED)) return; // :313
        p = cw.k;
        if (lead) ctl->leave_pos = p;
    } else {
        p = c.leave_pos;
    }
    const int neta = c.neta;
    if (threadIdx.x < R_) s_up[threadIdx.x] = threadIdx.x < neta ? U[(long long)threadIdx.x * ldu + p] : 0.0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int slot = dslot[r];
    const double base = slot >= 0 ? binv[(long long)p * ldb + slot]
                                  : (bcode[p] == -1 - r ? 1.0 : 0.0);
    double acc = 0.0;
    for (int t = 0; t < neta; ++t) acc = fma(s_up[t], W[(long long)t * ldw + r], acc);
    const double vr = base - acc;
    v[r] = vr;
    if (slot >= 0 && vc) vc[slot] = vr; // (compact copy: the row-wise pricing pass's coefficients)
}

// ---------------------------------------------------------------------------------
// k_fast_update: pivot() x4 (src/simplex.rs:262-265, :410-421) and, on the updated values, the
// per-workgroup first-pivot candidates of the NEXT iteration (src/simplex.rs:423-437).
// grid = DZG_NB_UPD workgroups of 256.  only_partials != 0: no update (initial state).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fast_update(DzgCtl *ctl, int only_partials, double *x,
                                                     double *xbar, double *z, double *zbar,
                                                     const double *__restrict__ dx,
                                                     const double *__restrict__ dz, int m, int q,
                                                     const int *__restrict__ nbcode, int col0,
                                                     int col1, int sharded, double *fpx_r,
                                                     int *fpx_k, double *fpx_h, double *fpz_r,
                                                     int *fpz_k, double *fpz_h,
                                                     const double *__restrict__ v,
                                                     double *__restrict__ U, long long ldu,
                                                     double *__restrict__ W, long long ldw,
                                                     double *__restrict__ binv, long long ldb,
                                                     int r0, int r1, int rowshard)
{
    // [r0, r1): the rows of x, xbar, U and Binv0 this device keeps -- [0, m) unless the basis side is
    // row-sharded (k_rowshard.hip), where dx_p comes from its owner's record (ctl->dxp) and the eta
    // row W_t = v, which every rank holds whole, is written for all m rows
    const DzgCtl c = *ctl; // one snapshot of the control block (scalar loads)
    if (c.status != DZG_RUNNING) return;
    const int p = c.leave_pos, r = c.enter_pos;
    if (!only_partials && c.del_last >= 0) { // compact column delete booked by fast_pivot_books
        const int ce = c.del_ce, last = c.del_last;
        for (int i = r0 + blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += gridDim.x * blockDim.x) {
            double *row = binv + (long long)i * ldb;
            if (ce != last) row[ce] = row[last];
            row[last] = 0.0;
        }
    }
    // eta of the pivot fast_pivot_books just booked (neta_new counts it): u = (dx - e_p)/dx_p,
    // w = v; if a slack entered, its row of W is structurally zero (its column became e_p)
    // the pivot's books (kept inside the previous launch) left the new eta / column counts in
    // *_next; this kernel works with them and commits them
    const int neta_new = only_partials ? c.neta : c.neta_next;
    if (!only_partials && blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->neta = c.neta_next;
        ctl->ncompact = c.ncompact_next;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->fp_count = (int)gridDim.x;
    const int teta = neta_new - 1;
    const int wzero = c.enter_code < 0 ? -1 - c.enter_code : -1;
    const double rdxp = only_partials ? 0.0 : 1.0 / (rowshard ? c.dxp : dx[p]);
    double *ut = U + (long long)(teta < 0 ? 0 : teta) * ldu;
    double *wt = W + (long long)(teta < 0 ? 0 : teta) * ldw;
    const double t = c.t, s = c.s, tbar = c.tbar, sbar = c.sbar;
    const int stride = gridDim.x * blockDim.x;
    // first-pivot candidates (src/simplex.rs:423-437) with their competition; a denominator
    // that is zero up to rounding (ybar in [-tau, tau]) is a candidate the reference may or may
    // not have -- with any ratio -- unless its numerator makes the ratio hopelessly negative
    const double tau = c.tau;
    DzgCand2 bx = dzg_cand2_none(), bz = dzg_cand2_none();
    if (!only_partials && rowshard) // (the rows of W_t outside this rank's share of x)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
            if (i < r0 || i >= r1) wt[i] = (i == wzero) ? 0.0 : v[i];
    for (int i = r0 + blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += stride) {
        double xi = x[i], xb = xbar[i];
        if (!only_partials) {
            const double d = dx[i];
            const double a = t * d, b = tbar * d;
            xi = (i == p) ? t : xi - a;
            xb = (i == p) ? tbar : xb - b;
            x[i] = xi;
            xbar[i] = xb;
            ut[i] = (i == p ? d - 1.0 : d) * rdxp;
            wt[i] = (i == wzero) ? 0.0 : v[i];
        }
        dzg_first_pivot_entry(bx, xi, xb, i, tau);
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < q; k += stride) {
        double zk = z[k], zb = zbar[k];
        if (!only_partials) {
            const double d = dz[k];
            const double a = s * d, b = sbar * d;
            zk = (k == r) ? s : zk - a;
            zb = (k == r) ? sbar : zb - b;
            z[k] = zk;
            zbar[k] = zb;
        }
        bool mine = true; // sharded: z is only maintained for slack positions and owned columns
        if (sharded) {
            const int code = nbcode[k];
            mine = code < 0 || (code >= col0 && code < col1);
        }
        if (mine) dzg_first_pivot_entry(bz, zk, zb, k, tau);
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
// Flush: Binv0[:, 0:k] -= U[:, 0:neta] * Wc,  Wc[t][c] = W[t][drow[c]].
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fast_gather_w(const DzgCtl *ctl, const double *__restrict__ W,
                                                       long long ldw, const int *__restrict__ drow,
                                                       double *__restrict__ Wc)
{
    // a flush folds a FULL eta file: one enqueued behind an iteration that did not pivot (the run
    // stopped, a barrier failed) is a no-op, so the flush falls after the same pivots whatever
    // happens in between -- results stay reproducible bit for bit
    if (ctl->neta < R_) return;
    const int k = ctl->ncompact, neta = ctl->neta;
    const int t = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int kpad = (k + 15) & ~15;
    if (c >= kpad) return;
    // rows neta..63 and columns k..kpad-1 are zero so the MFMA loop needs no masks on B
    Wc[(long long)t * ldw + c] = (t < neta && c < k) ? W[(long long)t * ldw + drow[c]] : 0.0;
}

// One wave owns a 16-row x 64-column strip of Binv0 (4 MFMA tiles side by side) and walks the
// eta index in steps of 4: D = C - A*B with v_mfma_f64_16x16x4_f64.  Operand lane maps
// (cdna_hip_programming.md section 3): A[i = l&15][kk = l>>4], B[kk = l>>4][j = l&15],
// C/D[row = (l>>4) + 4*reg][col = l&15].
template <bool PRELOAD>
__global__ __launch_bounds__(256) void k_fast_flush_mfma(const DzgCtl *ctl, int m,
                                                         double *__restrict__ binv, long long ldb,
                                                         const double *__restrict__ U, long long ldu,
                                                         const double *__restrict__ Wc,
                                                         long long ldw)
{
    const int neta = ctl->neta, k = ctl->ncompact;
    if (neta < R_ || k <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64;
    const int i0 = (blockIdx.y * 4 + wave) * 16;
    if (c0 >= k || i0 >= m) return;
    const int li = lane & 15, lk = lane >> 4;
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            acc[j][g] = (row < m && col < k) ? binv[(long long)row * ldb + col] : 0.0;
        }
    }
    const int arow = i0 + li;
    const int arowc = arow < m ? arow : 0;
    // The eta file is full (neta == R_ = 64): 16 steps of 4 etas, trip count known.  All operands
    // of the 64 MFMAs -- 16 values of U and 64 of Wc per lane -- are fetched BEFORE the first MFMA,
    // in one trip to L2 / HBM beside the tile of Binv0 itself: 161 us per flush at k = 4 060 against
    // 170 us fetched step by step (DZG_FLUSH_STEPS=1 selects that form for comparison).
    if (PRELOAD) {
        double av[R_ / 4], bv[R_ / 4][4];
#pragma unroll
        for (int s = 0; s < R_ / 4; ++s) {
            const int t = 4 * s + lk;
            av[s] = U[(long long)t * ldu + arowc];
            const double *wrow = Wc + (long long)t * ldw + c0 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[s][j] = wrow[16 * j];
        }
#pragma unroll
        for (int s = 0; s < R_ / 4; ++s) {
            const double a = arow < m ? -av[s] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s][j], acc[j], 0, 0, 0);
        }
    } else {
        for (int s = 0; s < R_ / 4; ++s) {
            const int t = 4 * s + lk;
            const double a = arow < m ? -U[(long long)t * ldu + arowc] : 0.0;
            const double *wrow = Wc + (long long)t * ldw + c0 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double b = wrow[16 * j];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            if (row < m && col < k) binv[(long long)row * ldb + col] = acc[j][g];
        }
    }
}

// The same product with the 64 x 64 tile of Wc the four waves of a workgroup share staged through
// LDS: fetched once per workgroup (one trip, beside the strip of U and the tile of Binv0) instead of
// once per wave -- 40 KB from the L2 per 8 KB of Binv0 became 16.  Same operands, same order of the
// eta steps: the same bits.
#define FLD 72 // LDS row stride in doubles (t -> t + 1 moves 16 banks on)
__global__ __launch_bounds__(256) void k_fast_flush_mfma_lds(const DzgCtl *ctl, int m,
                                                             double *__restrict__ binv, long long ldb,
                                                             const double *__restrict__ U, long long ldu,
                                                             const double *__restrict__ Wc, long long ldw)
{
    __shared__ double s_w[R_ * FLD];
    const int neta = ctl->neta, k = ctl->ncompact;
    if (neta < R_ || k <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64;
    const int i0 = (blockIdx.y * 4 + wave) * 16;
    if (c0 >= k || (int)blockIdx.y * 64 >= m) return; // (workgroup-uniform: the barrier below is safe)
    const int li = lane & 15, lk = lane >> 4;
    // one trip: the tile of Wc (row t = wave + 4 i, column c0 + lane: 512 contiguous bytes per wave
    // load; Wc is zero beyond the file and beyond round16(k), and its rows are ldw >= m + 64 long)
    double wreg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = Wc[(long long)(wave + 4 * i) * ldw + c0 + lane];
    const int arow = i0 + li;
    const int arowc = arow < m ? arow : 0;
    double av[R_ / 4];
#pragma unroll
    for (int s = 0; s < R_ / 4; ++s) av[s] = U[(long long)(4 * s + lk) * ldu + arowc];
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            acc[j][g] = (row < m && col < k) ? binv[(long long)row * ldb + col] : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s_w[(wave + 4 * i) * FLD + lane] = wreg[i];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < R_ / 4; ++s) {
        const double a = arow < m ? -av[s] : 0.0;
        const int t = 4 * s + lk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, s_w[t * FLD + 16 * j + li], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            if (row < m && col < k) binv[(long long)row * ldb + col] = acc[j][g];
        }
    }
}

// kcap: the compact width the flush's grid covered (the host's bound on k for the batch, k_hint).
// A wider inverse would have been flushed in part only: loud, not silent.
__global__ void k_fast_flush_done(DzgCtl *ctl, int kcap)
{
    if (ctl->neta >= R_) {
        if (ctl->ncompact > kcap && ctl->status == DZG_RUNNING) ctl->status = DZG_PANIC;
        ctl->neta = 0;
    }
}

__global__ __launch_bounds__(256) void k_fast_init(DzgCtl *ctl, int m, int q, int *dslot,
                                                   const int *__restrict__ nonbasis,
                                                   const int *__restrict__ var_col, int *plist,
                                                   int *pslot, int col0, int col1,
                                                   const long long *__restrict__ cptr, int *drow,
                                                   const int *__restrict__ basis, int *bcode,
                                                   int *nbcode, int *pcode, int *cpos)
{
    // single workgroup: the structural-position list must be built in position order
    for (int r = threadIdx.x; r < m; r += blockDim.x) {
        dslot[r] = -1;
        bcode[r] = var_col[basis[r]];
    }
    if (cpos)
        for (int j = threadIdx.x; j < col1 - col0; j += blockDim.x) cpos[j] = -1;
    __syncthreads();
    for (int k = threadIdx.x; k < q; k += blockDim.x) {
        const int code = var_col[nonbasis[k]];
        nbcode[k] = code;
        if (cpos && code >= col0 && code < col1) cpos[code - col0] = k;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < q; ++k) {
            const int code = var_col[nonbasis[k]];
            if (code >= col0 && code < col1) {
                plist[s] = k;
                pcode[s] = code;
                pslot[k] = s;
                ++s;
            } else {
                pslot[k] = -1;
            }
        }
        long long nnz = 0;
        if (cptr)
            for (int i = 0; i < s; ++i) {
                const int code = var_col[nonbasis[plist[i]]];
                nnz += cptr[code - col0 + 1] - cptr[code - col0];
            }
        ctl->nb_nnz = nnz;
        // dense columns of the inverse = rows whose slack is NOT basic.  None for the slack
        // basis Simplex::new builds; a caller-supplied basis (warm start) has some, and the
        // engine then builds Binv0 by a refactorisation before the first iteration.
        int kd = 0;
        for (int k = 0; k < q; ++k) {
            const int code = var_col[nonbasis[k]];
            if (code < 0) {
                drow[kd] = -1 - code;
                dslot[-1 - code] = kd;
                ++kd;
            }
        }
        ctl->ncompact = kd;
        ctl->neta = 0;
        ctl->nb_struct = s;
    }
}

// ---------------------------------------------------------------------------------
// Column sharding (one process per GPU).  Every rank runs the same O(m k) basis work on
// replicated x, dx, v, Binv; only the matrix, z and the pricing pass are split by column
// ownership.  Two exchanges per iteration, each one record per rank (layout: dantzig_amd.h).
// The merges below are deterministic (largest ratio, lowest GLOBAL position) and every rank
// sees the same records in the same order, so all ranks take identical decisions.
// ---------------------------------------------------------------------------------
// MODE 0: propose the first-pivot candidate of the z side (before status()).
// MODE 1: after pricing -- dual: propose the ratio-test candidate; primal: the owner of the
//         entering position publishes its z, zbar, dz.
// grid = 1 + ceil(m / 256): workgroup 0 writes the header, the others copy the column.
template <int MODE>
__global__ __launch_bounds__(256) void k_shard_propose(
    const DzgCtl *ctl, int m, const double *__restrict__ A, long long lda, int col0, int col1,
    const int *__restrict__ nbcode,
    const double *__restrict__ z, const double *__restrict__ zbar, const double *__restrict__ dz,
    const double *__restrict__ pr, const int *__restrict__ pk, const double *__restrict__ ph,
    int np, double *__restrict__ rec, int csc, int hdr_only)
{
    if (ctl->status != DZG_RUNNING) return;
    double ratio = 0.0, runner = -__builtin_inf();
    int pos = -1;
    bool want_column = true;
    if (MODE == 1 && ctl->kind == DZG_STEP_PRIMAL) {
        const int code = ctl->enter_code;
        if (code < 0 || (code >= col0 && code < col1)) {
            pos = ctl->enter_pos;
            ratio = 1.0;
        }
        want_column = false; // FTRAN already happened
    } else {
        // MODE 0: as many first-pivot candidates as the last update left (its grid size)
        const DzgCand2 c = reduce_partials(pr, pk, ph, MODE == 0 ? ctl->fp_count : np);
        pos = c.k;
        ratio = c.r;
        runner = c.h;
    }
    const int code = pos >= 0 ? nbcode[pos] : -1;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            rec[0] = ratio;
            rec[1] = (double)pos;
            rec[2] = pos >= 0 ? z[pos] : 0.0;
            rec[3] = pos >= 0 ? zbar[pos] : 0.0;
            rec[4] = (MODE == 1 && pos >= 0) ? dz[pos] : 0.0;
            rec[5] = (double)code;
            rec[6] = runner; // competition inside this rank (DzgCand2::h)
            rec[7] = 0.0;
        }
        return;
    }
    if (!want_column || code < 0 || hdr_only) return;
    const int i = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
    // sparse matrix: zero the slot here, k_shard_scatter_col then drops the stored entries in
    if (i < m) rec[8 + i] = csc ? 0.0 : A[(long long)(code - col0) * lda + i];
}

// CSC: scatter the proposed column's stored entries into the (zeroed) record.  One workgroup.
__global__ __launch_bounds__(256) void k_shard_scatter_col(const DzgCtl *ctl, int mode,
                                                           const long long *__restrict__ cptr,
                                                           const int *__restrict__ ridx,
                                                           const double *__restrict__ cval, int col0,
                                                           int col1, double *__restrict__ rec)
{
    if (ctl->status != DZG_RUNNING) return;
    if (mode == 1 && ctl->kind == DZG_STEP_PRIMAL) return; // no column in that record
    const int pos = (int)rec[1], code = (int)rec[5];
    if (pos < 0 || code < col0 || code >= col1) return;
    const long long e0 = cptr[code - col0], e1 = cptr[code - col0 + 1];
    for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const double val = cval[e];
        rec[8 + ridx[e]] = val;
    }
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
// every partial-candidate array starts as "no candidate, no competition": a launch that has nothing
// to price (no nonbasic position at all) leaves its array untouched, and the reduction behind it
// must not read what hipMalloc happened to hand over
__global__ __launch_bounds__(256) void k_fast_init_partials(double *r, int *k, double *h, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        r[i] = 0.0;
        k[i] = -1;
        h[i] = -__builtin_inf();
    }
}

void dzg_launch_fast_init(const DzgDev &d, hipStream_t st)
{
    {
        const int np = 4096; // (engine.hip allocates 4096 entries per array)
        double *rs[4] = {d.fpx_r, d.fpz_r, d.rx_r, d.rz_r}, *hs[4] = {d.fpx_h, d.fpz_h, d.rx_h, d.rz_h};
        int *ks[4] = {d.fpx_k, d.fpz_k, d.rx_k, d.rz_k};
        for (int a = 0; a < 4; ++a)
            hipLaunchKernelGGL(k_fast_init_partials, dim3(np / 256), dim3(256), 0, st, rs[a], ks[a], hs[a], np);
    }
    {
        const int r0 = d.rs ? d.rs_r0 : 0, rows = (d.rs ? d.rs_r1 : d.m) - r0;
        if (rows > 0)
            hipMemsetAsync(d.binv + (long long)r0 * d.ldb, 0, sizeof(double) * (size_t)rows * (size_t)d.ldb, st);
    }
    hipMemsetAsync(d.U, 0, sizeof(double) * (size_t)d.ldw * R_, st);
    hipMemsetAsync(d.W, 0, sizeof(double) * (size_t)d.ldw * R_, st);
    hipMemsetAsync(d.Wc, 0, sizeof(double) * (size_t)d.ldw * R_, st);
    hipMemsetAsync(d.ag, 0, sizeof(double) * ((size_t)d.m + 2), st);
    hipLaunchKernelGGL(k_fast_init, dim3(1), dim3(256), 0, st, d.ctl, d.m, d.q, d.dslot, d.nonbasis,
                       d.var_col, d.plist, d.pslot, d.col0, d.col1, d.csc ? d.cptr : nullptr, d.drow,
                       d.basis, d.bcode, d.nbcode, d.pcode, d.cpos);
}

void dzg_launch_fast_select_prep(const DzgDev &d, int mode, int nrz, const double *xrecv,
                                 hipStream_t st)
{
    // mode 0: status + primal prep          1: dual ratio test + prep          (one GPU)
    //      4: merge proposals + status + primal prep     5: merge second exchange + dual prep
#define SEL_ARGS d.ctl, d.m, d.A, d.lda, d.col0, xrecv, d.xstride, d.nbcode,                            \
                       d.fpx_r, d.fpx_k, d.fpx_h, d.fpz_r, d.fpz_k, d.fpz_h, d.rz_r, d.rz_k, d.rz_h, nrz,  \
                       d.W, d.ldw, d.drow, d.ag, d.beta, d.eps, d.world
    const dim3 grid(R_ + 1), block(256);
    switch (mode) {
    case 0: hipLaunchKernelGGL((k_fast_select_prep<0>), grid, block, 0, st, SEL_ARGS); break;
    case 1: hipLaunchKernelGGL((k_fast_select_prep<1>), grid, block, 0, st, SEL_ARGS); break;
    case 4: hipLaunchKernelGGL((k_fast_select_prep<4>), grid, block, 0, st, SEL_ARGS); break;
    default: hipLaunchKernelGGL((k_fast_select_prep<5>), grid, block, 0, st, SEL_ARGS); break;
    }
#undef SEL_ARGS
}

// need_kind == DZG_STEP_DUAL is the launch right before k_fast_update in every iteration: it also
// keeps the books of the pivot (fast_pivot_books), whatever the step kind.
void dzg_launch_fast_gemv(const DzgDev &d, int need_kind, const double *xrecv, hipStream_t st)
{
    const DzgPivotArgs pa = dzg_pivot_args(d);
#define GEMV_ARGS d.ctl, need_kind, d.m, d.binv, d.ldb, d.ag, d.U, d.ldw, d.beta, d.A, d.lda, d.col0, xrecv,  \
                  d.xstride, d.bcode, d.x, d.xbar, d.dx, d.rx_r, d.rx_k, d.rx_h, pa,                         \
                  (d.ftran_variant >= 0 ? (d.ftran_variant ? 0 : 0x7fffffff) : d.ftran_nt_k)
    if (need_kind == DZG_STEP_DUAL)
        hipLaunchKernelGGL((k_fast_gemv<true>), dim3(DZG_NB_GEMV), dim3(256), 0, st, GEMV_ARGS);
    else
        hipLaunchKernelGGL((k_fast_gemv<false>), dim3(DZG_NB_GEMV), dim3(256), 0, st, GEMV_ARGS);
#undef GEMV_ARGS
}

void dzg_launch_fast_btran(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_fast_btran, dim3((d.m + 255) / 256), dim3(256), 0, st, d.ctl, d.m, d.binv,
                       d.ldb, d.dslot, d.bcode, d.U, d.ldw, d.W, d.ldw, d.rx_r, d.rx_k, d.rx_h,
                       d.v, d.vc);
}

void dzg_launch_fast_update(const DzgDev &d, int only_partials, hipStream_t st)
{
    hipLaunchKernelGGL(k_fast_update, dim3(DZG_NB_UPD), dim3(256), 0, st, d.ctl, only_partials, d.x,
                       d.xbar, d.z, d.zbar, d.dx, d.dz, d.m, d.q, d.nbcode, d.col0, d.col1,
                       d.world > 1 ? 1 : 0, d.fpx_r, d.fpx_k, d.fpx_h, d.fpz_r, d.fpz_k, d.fpz_h, d.v, d.U,
                       d.ldw, d.W, d.ldw, d.binv, d.ldb, d.rs ? d.rs_r0 : 0, d.rs ? d.rs_r1 : d.m, d.rs);
}

void dzg_launch_fast_flush(const DzgDev &d, hipStream_t st)
{
    // ncompact <= k_hint <= m (the host's bound for the batch in flight); the kernels mask by ctl
    const int kmax = d.k_hint > 0 && d.k_hint < d.m ? d.k_hint : d.m;
    hipLaunchKernelGGL(k_fast_gather_w, dim3((kmax + 15 + 255) / 256, R_), dim3(256), 0, st, d.ctl,
                       d.W, d.ldw, d.drow, d.Wc);
    // row-sharded basis side: this rank's rows only (a multiple of 16 rows from the top, so the
    // MFMA tiles are the single-GPU solve's tiles: the same bits)
    const int r0 = d.rs ? d.rs_r0 : 0, rows = (d.rs ? d.rs_r1 : d.m) - r0;
    double *binv = d.binv + (long long)r0 * d.ldb;
    const double *U = d.U + r0;
    static const bool steps = std::getenv("DZG_FLUSH_STEPS") != nullptr; // (A/B switch, tools)
    static const bool no_lds = std::getenv("DZG_FLUSH_NO_LDS") != nullptr; // (A/B switch, tools)
    if (rows <= 0) {
    } else if (!steps && !no_lds)
        hipLaunchKernelGGL(k_fast_flush_mfma_lds, dim3((kmax + 63) / 64, (rows + 63) / 64), dim3(256), 0, st,
                           d.ctl, rows, binv, d.ldb, U, d.ldw, d.Wc, d.ldw);
    else if (steps)
        hipLaunchKernelGGL(k_fast_flush_mfma<false>, dim3((kmax + 63) / 64, (rows + 63) / 64), dim3(256), 0, st,
                           d.ctl, rows, binv, d.ldb, U, d.ldw, d.Wc, d.ldw);
    else
        hipLaunchKernelGGL(k_fast_flush_mfma<true>, dim3((kmax + 63) / 64, (rows + 63) / 64), dim3(256), 0, st,
                           d.ctl, rows, binv, d.ldb, U, d.ldw, d.Wc, d.ldw);
    hipLaunchKernelGGL(k_fast_flush_done, dim3(1), dim3(1), 0, st, d.ctl, ((kmax + 63) / 64) * 64);
}

void dzg_launch_shard_propose(const DzgDev &d, int mode, int nrz, double *xsend, hipStream_t st)
{
    const int hdr_only = d.xstride <= 8; // replicated matrix: the column is read locally
    const dim3 grid(hdr_only ? 1 : 1 + (d.m + 255) / 256);
    if (mode == 0)
        hipLaunchKernelGGL((k_shard_propose<0>), grid, dim3(256), 0, st, d.ctl, d.m, d.A, d.lda,
                           d.col0, d.col1, d.nbcode, d.z, d.zbar, d.dz, d.fpz_r,
                           d.fpz_k, d.fpz_h, DZG_NB_UPD, xsend, d.csc, hdr_only);
    else
        hipLaunchKernelGGL((k_shard_propose<1>), grid, dim3(256), 0, st, d.ctl, d.m, d.A, d.lda,
                           d.col0, d.col1, d.nbcode, d.z, d.zbar, d.dz, d.rz_r, d.rz_k,
                           d.rz_h, nrz, xsend, d.csc, hdr_only);
    if (d.csc)
        hipLaunchKernelGGL(k_shard_scatter_col, dim3(1), dim3(256), 0, st, d.ctl, mode, d.cptr,
                           d.ridx, d.cval, d.col0, d.col1, xsend);
}

// Lockstep harness (all ranks in one process on one GPU): recv_dst[src] = send_src for every pair.
// ptrs = [send_0..send_{P-1} | recv1_0.. | recv2_0..]; which = 1 or 2.  grid (blocks over a record, P*P)
__global__ __launch_bounds__(256) void k_lockstep_allgather(double *const *ptrs, int world, int which,
                                                            long long xstride)
{
    const int pair = blockIdx.y, dst = pair / world, src = pair % world;
    const double *from = ptrs[src];
    double *to = ptrs[(long long)which * world + dst] + (long long)src * xstride;
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < xstride; i += (long long)gridDim.x * blockDim.x)
        to[i] = from[i];
}

void dzg_launch_lockstep_allgather(double *const *ptrs, int world, int which, long long xstride,
                                   hipStream_t st)
{
    int bx = (int)((xstride + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_lockstep_allgather, dim3(bx, world * world), dim3(256), 0, st, ptrs, world,
                       which, xstride);
}

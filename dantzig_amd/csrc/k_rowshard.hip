// k_rowshard.hip -- column sharding with the BASIS SIDE sharded as well (opts.shard_rows).
//
// Plain column sharding (k_fast.hip, DESIGN.md "Multi-GPU") divides the pricing pass by the number
// of ranks and replicates everything on the x side: every rank streams the whole compact inverse
// for FTRAN (8 m k bytes) and folds the whole eta file into it.  Deep in a solve that is what an
// iteration costs (the pricing pass sheds what FTRAN takes up), so more ranks buy nothing there.
// Here rank r also owns the ROWS [r S, (r + 1) S) of the x side -- x, xbar, dx, Binv0, the eta
// columns U -- and FTRAN, the x-side ratio test, the eta flush and the update touch those rows only.
//
// What does not shard is one row: BTRAN is "row p of the inverse", and row p lives on one rank.
// The ranks therefore ship candidates WITH their rows: every x-side candidate a rank proposes --
// its first-pivot candidate in exchange 1, its ratio-test candidate of a primal step in exchange 2
// -- travels with row p of Binv0 and the 64 values U_t[p] (record layout: include/dantzig_amd.h).
// After the merge every rank evaluates v = row p of the inverse from the winner's record with the
// arithmetic the row's owner would use (k_fast_btran's), and in a dual step dx_p = (row p) . a_j
// with fast_gemv_row's, so x_p / dx_p needs no third exchange.  Still two exchanges per iteration:
//
//   phase 1   propose: z-side first pivot (+ column)  |  x-side first pivot of the own rows + row p
//   exchange 1
//   phase 2   merge both sides, status();  primal: beta, FTRAN on the own rows, ratio candidates
//             -> propose x-side ratio candidate + row p;   dual: v from the winner's row, pricing
//             of the own columns -> propose z-side ratio candidate (+ column)
//   exchange 2
//   phase 3   primal: merge -> p, v from the winner's row, pricing, dz_r of the entering column by
//             EVERY rank (one wave, the pricing pass's own summation order: the owner's bits);
//             dual: merge -> entering column, beta, FTRAN on the own rows, dx_p from the row
//             exchange 1 delivered;   then the pivot's books (replicated) and the update
//
// Every formula is the function the other FAST kernels call (fast_rows.h, fast_decide.h): a
// row-sharded solve takes, bit for bit, the pivots of the single-GPU solve (tests/test_sharded.py).
#include "common.h"
#include "fast_decide.h"
#include "fast_rows.h"

#define RS_HDR DZG_RS_HDR
#define RS_COL DZG_RS_COL

// offset of the x-side candidate in a record header: same relative layout as the z side
// (+0 ratio, +1 position, +6 runner-up), so one merge serves both
#define RS_X 8

__device__ __forceinline__ int rs_merge(const double *__restrict__ xrecv, long long xstride, int world,
                                        int off, DzgCand2 &win)
{
    int w = -1;
    win = dzg_cand2_none();
    for (int r = 0; r < world; ++r) {
        const double *rec = xrecv + (long long)r * xstride + off;
        DzgCand2 c;
        c.r = rec[0];
        c.k = (int)rec[1];
        c.h = rec[6];
        if (c.k < 0 || c.r != c.r || c.k == win.k) { // (replicated slack positions: one candidate)
            if (c.h > win.h) win.h = c.h;
            continue;
        }
        win = dzg_better2(win, c);
        if (win.k == c.k) w = r;
    }
    return w;
}

// the entering column: the winner's record (partitioned storage) or the local matrix (replicated)
__device__ __forceinline__ const double *rs_col(const DzgDev &d, int code,
                                                const double *__restrict__ xrecv, int src)
{
    if (code < 0) return nullptr;
    if (d.rs_mcol > 0) return xrecv + (long long)src * d.xstride + RS_COL;
    return d.A + (long long)(code - d.col0) * d.lda;
}

__device__ __forceinline__ const double *rs_row(const DzgDev &d, const double *__restrict__ xrecv, int src)
{
    return xrecv + (long long)src * d.xstride + RS_COL + d.rs_mcol;
}

// ---------------------------------------------------------------------------------
// k_rs_propose<MODE>: this rank's record.
//   MODE 0 (phase 1): z-side first-pivot candidate (src/simplex.rs:275) with its column, x-side
//           first-pivot candidate of the own rows (:276) with row p of Binv0 and U_t[p];
//   MODE 1 (end of phase 2): dual step -- z-side ratio-test candidate (:324) with its column;
//           primal step -- x-side ratio-test candidate of the own rows (:313) with x_p, xbar_p,
//           dx_p, row p and U_t[p].
// grid = 1 + ceil(mcol / 256) + ceil(krow / 256): header | column | row.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_rs_propose(const DzgDev d, int nrz, double *__restrict__ rec)
{
    const DzgCtl *ctl = d.ctl;
    if (ctl->status != DZG_RUNNING) return;
    const int kind = ctl->kind, neta = ctl->neta;
    DzgCand2 zc = dzg_cand2_none(), xc = dzg_cand2_none();
    if (MODE == 0) {
        zc = reduce_partials(d.fpz_r, d.fpz_k, d.fpz_h, ctl->fp_count);
        xc = reduce_partials(d.fpx_r, d.fpx_k, d.fpx_h, ctl->fp_count);
    } else if (kind == DZG_STEP_DUAL) {
        zc = reduce_partials(d.rz_r, d.rz_k, d.rz_h, nrz);
    } else {
        xc = reduce_partials(d.rx_r, d.rx_k, d.rx_h, DZG_NB_GEMV);
    }
    const int zpos = zc.k, p = xc.k;
    const int code = zpos >= 0 ? d.nbcode[zpos] : -1;
    const int ncolb = (int)((d.rs_mcol + 255) / 256);
    const int b = blockIdx.x;
    if (b == 0) {
        if (threadIdx.x == 0) {
            rec[0] = zc.r;
            rec[1] = (double)zpos;
            rec[2] = zpos >= 0 ? d.z[zpos] : 0.0;
            rec[3] = zpos >= 0 ? d.zbar[zpos] : 0.0;
            rec[4] = (MODE == 1 && zpos >= 0) ? d.dz[zpos] : 0.0;
            rec[5] = (double)code;
            rec[6] = zc.h;
            rec[7] = 0.0;
            rec[RS_X + 0] = xc.r;
            rec[RS_X + 1] = (double)p;
            rec[RS_X + 2] = p >= 0 ? d.x[p] : 0.0;
            rec[RS_X + 3] = p >= 0 ? d.xbar[p] : 0.0;
            rec[RS_X + 4] = (MODE == 1 && p >= 0) ? d.dx[p] : 0.0;
            rec[RS_X + 5] = 0.0;
            rec[RS_X + 6] = xc.h;
            rec[RS_X + 7] = 0.0;
        }
        if (threadIdx.x < R_)
            rec[RS_HDR + threadIdx.x] = (p >= 0 && (int)threadIdx.x < neta)
                                            ? d.U[(long long)threadIdx.x * d.ldw + p] : 0.0;
        return;
    }
    if (b <= ncolb) { // the z-side candidate's column (an owned column: this rank holds it)
        if (code < 0) return;
        const int i = (b - 1) * 256 + (int)threadIdx.x;
        if (i < d.m) rec[RS_COL + i] = d.A[(long long)(code - d.col0) * d.lda + i];
        return;
    }
    if (p < 0) return; // row p of Binv0: the columns beyond k are kept zero, the record's pad too
    const int c = (b - 1 - ncolb) * 256 + (int)threadIdx.x;
    const long long krow = d.xstride - RS_COL - d.rs_mcol;
    if (c < krow) rec[RS_COL + d.rs_mcol + c] = c < d.ldb ? d.binv[(long long)p * d.ldb + c] : 0.0;
}

// v = row p of the inverse on row r, from the owner's record: k_fast_btran's arithmetic
__device__ __forceinline__ void rs_btran_row(const DzgDev &d, int neta, int p, int lcode,
                                             const double *__restrict__ base,
                                             const double *__restrict__ up, int r)
{
    const int slot = d.dslot[r];
    const double b0 = slot >= 0 ? base[slot] : (lcode == -1 - r ? 1.0 : 0.0);
    double acc = 0.0;
    for (int t = 0; t < neta; ++t) acc = fma(up[t], d.W[(long long)t * d.ldw + r], acc);
    const double vr = b0 - acc;
    d.v[r] = vr;
    if (slot >= 0 && d.vc) d.vc[slot] = vr;
    (void)p;
}

// FTRAN preparation for the entering variable (k_fast_select_prep's): workgroup t < neta forms
// beta_t = W_t . a_j, workgroup R_ gathers a_j to compact coordinates
__device__ __forceinline__ void rs_ftran_prep(const DzgDev &d, int code, const double *__restrict__ a,
                                              int neta, int k)
{
    const int b = blockIdx.x;
    if (b < R_) {
        if (b >= neta) return;
        const double *wt = d.W + (long long)b * d.ldw;
        if (code < 0) {
            if (threadIdx.x == 0) d.beta[b] = wt[-1 - code];
            return;
        }
        const double acc = fast_beta_dot(wt, a, d.m);
        if (threadIdx.x == 0) d.beta[b] = acc;
    } else {
        // the gather to compact coordinates, spread over the workgroups behind the beta ones (one
        // workgroup took 20 us for k = 16 384: each entry is two dependent loads)
        const int nb = (int)gridDim.x - R_, stride = nb * (int)blockDim.x;
        const int rr = -1 - code;
        for (int c = (b - R_) * (int)blockDim.x + (int)threadIdx.x; c < k; c += stride)
            d.ag[c] = code < 0 ? ((d.drow[c] == rr) ? 1.0 : 0.0) : a[d.drow[c]];
        if (b == R_ && threadIdx.x == 0 && (k & 1)) d.ag[k] = 0.0;
    }
}

// ---------------------------------------------------------------------------------
// k_rs_select<MODE>: the head of phase 2 (MODE 0, records of exchange 1) and of phase 3 (MODE 1,
// exchange 2).  Every workgroup merges the records itself -- all ranks see the same records in the
// same order and apply the same rule (largest ratio, lowest position: src/simplex.rs:432-435,
// :456-459), so they take the same decision; the lead lane records it.
//   MODE 0: status() (:274-306).  Primal: FTRAN preparation.  Dual: v from the x winner's row.
//   MODE 1: dual: ratio test on the z side (:324-325), FTRAN preparation; primal: ratio test on the
//           x side (:313), v from the winner's row.
// grid = R_ + 1 + ceil(m / 256): beta | gather | v rows.
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_rs_select(const DzgDev d, const double *__restrict__ xrecv)
{
    DzgCtl *ctl = d.ctl;
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int neta = c.neta, k = c.ncompact;
    int prep_code = 0, row_src = -1, p = -1;
    const double *a = nullptr;
    bool prep = false, rows = false;
    if (MODE == 0) {
        DzgCand2 cj, ci;
        const int wz = rs_merge(xrecv, d.xstride, d.world, 0, cj);
        const int wx = rs_merge(xrecv, d.xstride, d.world, RS_X, ci);
        int kind;
        if (!fast_status(ctl, c, lead, cj, ci, d.eps, d.m, true, kind)) return;
        if (kind == DZG_STEP_PRIMAL) {
            const double *rec = xrecv + (long long)wz * d.xstride;
            prep_code = (int)rec[5];
            if (lead) {
                ctl->enter_src = wz;
                ctl->enter_code = prep_code;
                ctl->zr = rec[2];
                ctl->zbar_r = rec[3];
            }
            a = rs_col(d, prep_code, xrecv, wz);
            prep = true;
        } else {
            const double *rec = xrecv + (long long)wx * d.xstride;
            p = ci.k;
            row_src = wx;
            if (lead) {
                ctl->leave_src = wx;
                ctl->xp = rec[RS_X + 2];
                ctl->xbp = rec[RS_X + 3];
                ctl->leave_code = d.bcode[p];
            }
            rows = true;
        }
    } else if (c.kind == DZG_STEP_DUAL) {
        DzgCand2 cw;
        const int w = rs_merge(xrecv, d.xstride, d.world, 0, cw);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) return; // src/simplex.rs:325
        const double *rec = xrecv + (long long)w * d.xstride;
        prep_code = (int)rec[5];
        if (lead) {
            ctl->enter_pos = cw.k;
            ctl->enter_src = w;
            ctl->enter_code = prep_code;
            ctl->zr = rec[2];
            ctl->zbar_r = rec[3];
            ctl->dz_r = rec[4];
            ctl->use_record = 1;
        }
        a = rs_col(d, prep_code, xrecv, w);
        prep = true;
    } else {
        DzgCand2 cw;
        const int w = rs_merge(xrecv, d.xstride, d.world, RS_X, cw);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_UNBOUNDED)) return; // src/simplex.rs:313
        const double *rec = xrecv + (long long)w * d.xstride;
        p = cw.k;
        row_src = w;
        if (lead) {
            ctl->leave_pos = p;
            ctl->leave_src = w;
            ctl->xp = rec[RS_X + 2];
            ctl->xbp = rec[RS_X + 3];
            ctl->dxp = rec[RS_X + 4];
            ctl->leave_code = d.bcode[p];
        }
        rows = true;
    }
    if (prep) { // (every workgroup takes part: beta | gather)
        rs_ftran_prep(d, prep_code, a, neta, k);
        return;
    }
    if (rows && (int)blockIdx.x > R_) {
        const int r = ((int)blockIdx.x - R_ - 1) * 256 + (int)threadIdx.x;
        if (r < d.m) {
            const double *recw = xrecv + (long long)row_src * d.xstride;
            rs_btran_row(d, neta, p, d.bcode[p], rs_row(d, xrecv, row_src), recw + RS_HDR, r);
        }
    }
}

// ---------------------------------------------------------------------------------
// k_rs_gemv<KIND>: dx = Binv a_j on this rank's rows (k_fast_gemv's rows: fast_gemv_row +
// fast_gemv_unit).  KIND = PRIMAL (phase 2): also the ratio-test candidates of those rows
// (src/simplex.rs:439-461).  KIND = DUAL (phase 3): also dx_p, by one wave of workgroup 0, from the
// row of Binv0 that exchange 1 delivered with the leaving position -- the owner's arithmetic on
// the owner's numbers, so every rank holds the owner's bits.
// grid = DZG_NB_GEMV workgroups of 256.
// ---------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ void rs_gemv_rows(const DzgDev &d, const DzgCtl &c, int need_kind, int k,
                                             int neta, int code, const double *__restrict__ acolp,
                                             DzgCand2 &best)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int k2 = (k + 1) & ~1;
    const double mu = c.mu, tau = c.tau;
    const long long own = d.rs_r1 - d.rs_r0;
    const bool nt = d.ftran_variant >= 0 ? d.ftran_variant != 0 : 8.0 * (double)own * (double)k > 384e6;
    for (int i0 = d.rs_r0 + wave_global * RPW; i0 < d.rs_r1; i0 += nwaves * RPW) {
        const int i = i0 + grp;
        const int ii = i < d.rs_r1 ? i : d.m; // (rows beyond the share: zero, the lanes still fold)
        // (a rank's share of the inverse is 8 (m / P) k bytes: it streams past the caches from the
        // width on at which THAT outgrows the Infinity Cache)
        double acc = (LPR == 64 && nt)
                         ? fast_gemv_row<LPR, 2>(ii, d.m, k2, neta, d.binv, d.ldb, d.ag, d.U, d.ldw, d.beta, sub)
                         : fast_gemv_row<LPR, 0>(ii, d.m, k2, neta, d.binv, d.ldb, d.ag, d.U, d.ldw, d.beta, sub);
        if (ii < d.m && sub == 0) {
            acc = fast_gemv_unit(acc, d.bcode[i], code, acolp);
            d.dx[i] = acc;
            if (need_kind == DZG_STEP_PRIMAL) {
                const double xi = d.x[i], scaled = mu * d.xbar[i];
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

template <int KIND>
__global__ __launch_bounds__(256) void k_rs_gemv(const DzgDev d, const double *__restrict__ xrecv1,
                                                 const double *__restrict__ xrecv2)
{
    DzgCtl *ctl = d.ctl;
    const DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING || c.kind != KIND) return;
    const int k = c.ncompact, neta = c.neta, code = c.enter_code;
    // (primal: the entering column came with exchange 1; dual: with exchange 2)
    const double *acolp = rs_col(d, code, KIND == DZG_STEP_PRIMAL ? xrecv1 : xrecv2, c.enter_src);
    DzgCand2 best = dzg_cand2_none();
    if (k > 512)
        rs_gemv_rows<64>(d, c, KIND, k, neta, code, acolp, best);
    else
        rs_gemv_rows<16>(d, c, KIND, k, neta, code, acolp, best);
    if (KIND == DZG_STEP_PRIMAL) {
        best = dzg_block_best2(best);
        if (threadIdx.x == 0) {
            d.rx_r[blockIdx.x] = best.r;
            d.rx_k[blockIdx.x] = best.k;
            d.rx_h[blockIdx.x] = best.h;
        }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) { // dx_p from the row exchange 1 delivered
        const int lane = threadIdx.x, p = c.leave_pos;
        const double *rec = xrecv1 + (long long)c.leave_src * d.xstride;
        const double *base = rs_row(d, xrecv1, c.leave_src), *up = rec + RS_HDR;
        const int k2 = (k + 1) & ~1;
        double acc;
        // (row 0 of a one-row matrix whose row is `base` and whose eta column entries are up[t])
        if (k > 512)
            acc = fast_gemv_row<64>(0, 1, k2, neta, base, 0, d.ag, up, 1, d.beta, lane);
        else
            acc = fast_gemv_row<16>(lane < 16 ? 0 : 1, 1, k2, neta, base, 0, d.ag, up, 1, d.beta, lane % 16);
        if (lane == 0) ctl->dxp = fast_gemv_unit(acc, d.bcode[p], code, acolp);
    }
}

// ---------------------------------------------------------------------------------
// dz of ONE structural column as this iteration's pricing pass computes it, by one wave.  A primal
// step prices after its second exchange, and every rank needs dz_r of the entering column for the
// step lengths s, sbar (src/simplex.rs:258-260) -- only its owner's pricing pass has it.  Both
// passes sum a column in an order that depends on the column alone (k_price_kernels.h):
//   row-wise (k < rows_T): G = min(32, ceil((k + 1) / 16)) groups, group g the rows c = g, g + G,
//     ... of the compact numbering (the leaving slack's own row last, coefficient 1), one fma
//     chain each; the G partial sums added in group order;
//   column-wise (k_price_tree): lane l the rows 128 t + 2 l, + 1 of every tile t, one fma chain, the
//     64 partial sums folded by an xor-shuffle tree.
// ---------------------------------------------------------------------------------
// (called by ALL 256 threads of the workgroup; the result is valid in the lanes of wave 0)
#define RS_TILE 2048
__device__ __forceinline__ double rs_price_one(const DzgDev &d, const double *__restrict__ a, int k,
                                               int lcode, int lane)
{
    if (d.At && d.rows_T > 0 && k < d.rows_T) {
        // The chains are fma after fma in a fixed order; what need not wait in line are the loads
        // (coefficient, row index, then the column's entry: two dependent trips per term, 34 us for
        // k = 16 384 when one wave did it all).  The whole workgroup stages a tile of (coefficient,
        // entry) pairs in LDS, the G lanes of wave 0 then run their chains out of LDS.
        __shared__ double s_cf[RS_TILE], s_av[RS_TILE];
        int G = (k + 1 + DZG_PR_BATCH - 1) / DZG_PR_BATCH;
        G = G > DZG_PR_GMAX ? DZG_PR_GMAX : G;
        const int nrows = k + (lcode < 0 ? 1 : 0);
        const int tid = threadIdx.x;
        double acc = 0.0;
        for (int c0 = 0; c0 < nrows; c0 += RS_TILE) { // (block-uniform trip count)
            __syncthreads();
            for (int i = tid; i < RS_TILE && c0 + i < nrows; i += blockDim.x) {
                const int c = c0 + i;
                const int row = c < k ? d.drow[c] : -1 - lcode;
                s_cf[i] = c < k ? d.vc[c] : 1.0;
                s_av[i] = a[row];
            }
            __syncthreads();
            if (tid < G) {
                // this lane's terms inside the tile: c = lane, lane + G, ... (global numbering)
                int c = c0 + ((tid - c0 % G) % G + G) % G;
                const int cend = c0 + RS_TILE < nrows ? c0 + RS_TILE : nrows;
                for (; c < cend; c += G) acc = fma(s_cf[c - c0], s_av[c - c0], acc);
            }
        }
        double sum = 0.0;
        for (int g = 0; g < G; ++g) sum = sum + __shfl(acc, g, DZG_WAVE);
        return -sum;
    }
    const int m = d.m, ntiles = (m + 127) / 128;
    double acc = 0.0;
    for (int t = 0; t < (threadIdx.x < 64 ? ntiles : 0); ++t) {
        const int row = t * 128 + 2 * lane;
        const bool inside = row < m; // (row m of a column is zero: the matrix's padding, the record's)
        const double ax = inside ? a[row] : 0.0, ay = inside ? a[row + 1] : 0.0;
        const int rv = inside ? row : 0;
        const double vx = d.v[rv], vy = d.v[rv + 1]; // (v carries two zero pads)
        acc = fma(ax, vx, acc);
        acc = fma(ay, vy, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, DZG_WAVE);
    return -acc;
}

// ---------------------------------------------------------------------------------
// k_rs_books: the pivot's books (fast_rows.h) on every rank alike, from scalars every rank holds:
// x_p, xbar_p, dx_p from the leaving row's owner, z_r, zbar_r, dz_r from the entering column's.
// One workgroup; the launch before k_fast_update, which commits the counts.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rs_books(const DzgDev d, const DzgPivotArgs pa,
                                                  const double *__restrict__ xrecv1)
{
    __shared__ double s_dzr;
    DzgCtl *ctl = d.ctl;
    const DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    double dzr = c.dz_r;
    if (c.kind == DZG_STEP_PRIMAL) { // (a dual step's came with the winner's record)
        const int code = c.enter_code;
        double v0;
        if (code < 0) // a slack position: every rank priced the unit column itself
            v0 = d.dz[c.enter_pos];
        else          // (all 256 threads: the terms are staged through LDS; valid in wave 0)
            v0 = rs_price_one(d, rs_col(d, code, xrecv1, c.enter_src), c.ncompact, c.leave_code,
                              (int)(threadIdx.x & 63));
        if (threadIdx.x == 0) s_dzr = v0;
        __syncthreads();
        dzr = s_dzr;
    }
    DzgPivotScalars ps;
    ps.ok = 1;
    if (threadIdx.x == 0) {
        ps = fast_pivot_scalars(c.xp, c.xbp, c.dxp, c.zr, c.zbar_r, dzr, c.neta, c.max_pivot_err);
        if (c.kind == DZG_STEP_PRIMAL) {
            ctl->dz_r = dzr;
            ctl->use_record = 1;
        }
    }
    fast_pivot_books_s(ctl, c, pa, ps, 0);
}

// x, xbar of every rank's rows into every rank's arrays (lockstep harness: all ranks in one
// process).  ptrs = [x_0 .. x_{P-1} | xbar_0 .. xbar_{P-1}];  grid (blocks over a slice, P)
__global__ __launch_bounds__(256) void k_rs_lockstep_gather(double *const *ptrs, int world, int m,
                                                            int slice)
{
    const int src = blockIdx.y;
    const int r0 = src * slice < m ? src * slice : m, r1 = r0 + slice < m ? r0 + slice : m;
    for (int i = r0 + blockIdx.x * blockDim.x + threadIdx.x; i < r1; i += gridDim.x * blockDim.x) {
        const double xv = ptrs[src][i], xb = ptrs[world + src][i];
        for (int dst = 0; dst < world; ++dst) {
            if (dst == src) continue;
            ptrs[dst][i] = xv;
            ptrs[world + dst][i] = xb;
        }
    }
}

// a rank's slice of x / xbar into a send buffer of `slice` doubles each (zero beyond the rows), and
// back from the gathered slices [rank][x slice | xbar slice] into x / xbar
__global__ __launch_bounds__(256) void k_rs_pack(const DzgDev d, int slice, double *__restrict__ send)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slice) return;
    const int r = d.rs_r0 + i;
    send[i] = r < d.rs_r1 ? d.x[r] : 0.0;
    send[slice + i] = r < d.rs_r1 ? d.xbar[r] : 0.0;
}
__global__ __launch_bounds__(256) void k_rs_unpack(const DzgDev d, int slice, const double *__restrict__ recv)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= d.m) return;
    const int src = r / slice, i = r - src * slice;
    d.x[r] = recv[(long long)src * 2 * slice + i];
    d.xbar[r] = recv[(long long)src * 2 * slice + slice + i];
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
void dzg_launch_rs_propose(const DzgDev &d, int mode, int nrz, double *xsend, hipStream_t st)
{
    const long long krow = d.xstride - RS_COL - d.rs_mcol;
    const dim3 grid((unsigned)(1 + (d.rs_mcol + 255) / 256 + (krow + 255) / 256));
    if (mode == 0)
        hipLaunchKernelGGL((k_rs_propose<0>), grid, dim3(256), 0, st, d, nrz, xsend);
    else
        hipLaunchKernelGGL((k_rs_propose<1>), grid, dim3(256), 0, st, d, nrz, xsend);
}

void dzg_launch_rs_select(const DzgDev &d, int mode, const double *xrecv, hipStream_t st)
{
    const dim3 grid((unsigned)(R_ + 1 + (d.m + 255) / 256));
    if (mode == 0)
        hipLaunchKernelGGL((k_rs_select<0>), grid, dim3(256), 0, st, d, xrecv);
    else
        hipLaunchKernelGGL((k_rs_select<1>), grid, dim3(256), 0, st, d, xrecv);
}

void dzg_launch_rs_gemv(const DzgDev &d, int kind, const double *xrecv1, const double *xrecv2,
                        hipStream_t st)
{
    if (kind == DZG_STEP_PRIMAL)
        hipLaunchKernelGGL((k_rs_gemv<DZG_STEP_PRIMAL>), dim3(DZG_NB_GEMV), dim3(256), 0, st, d, xrecv1, xrecv2);
    else
        hipLaunchKernelGGL((k_rs_gemv<DZG_STEP_DUAL>), dim3(DZG_NB_GEMV), dim3(256), 0, st, d, xrecv1, xrecv2);
}

void dzg_launch_rs_books(const DzgDev &d, const double *xrecv1, hipStream_t st)
{
    hipLaunchKernelGGL(k_rs_books, dim3(1), dim3(256), 0, st, d, dzg_pivot_args(d), xrecv1);
}

void dzg_launch_rs_lockstep_gather(double *const *ptrs, int world, int m, int slice, hipStream_t st)
{
    if (m <= 0 || slice <= 0) return;
    int bx = (slice + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_rs_lockstep_gather, dim3(bx, world), dim3(256), 0, st, ptrs, world, m, slice);
}

void dzg_launch_rs_pack(const DzgDev &d, int slice, double *send, hipStream_t st)
{
    hipLaunchKernelGGL(k_rs_pack, dim3((slice + 255) / 256), dim3(256), 0, st, d, slice, send);
}

void dzg_launch_rs_unpack(const DzgDev &d, int slice, const double *recv, hipStream_t st)
{
    hipLaunchKernelGGL(k_rs_unpack, dim3((d.m + 255) / 256), dim3(256), 0, st, d, slice, recv);
}

// k_chain.hip -- the FAST iteration of the dense inverse in THREE launches on one GPU:
//
//      k_chain_pre   status(); a primal step: beta, FTRAN, ratio test;   then BTRAN's row
//      pricing       (k_price.hip, unchanged: the only kernel that touches the matrix)
//      k_chain_post  a dual step: ratio test, beta, FTRAN;   the pivot's books;   the update
//
// The seven-launch form of the same iteration (k_fast.hip) pays a kernel boundary for every
// dependent phase.  Here the phases of one side of the pricing pass share a launch of one
// workgroup per CU, and two things keep the number of device-wide synchronisations below the number
// of phases:
//
//   * Every decision is taken by EVERY workgroup from the same partial results (a few KB), so a
//     decision needs no broadcast; the lead lane of workgroup 0 only records it in the control
//     block for the launches that follow.
//   * A workgroup owns the same rows in FTRAN, BTRAN and the update and the same columns of z, and
//     nothing it reads inside a launch is rewritten by another workgroup of that launch: the six
//     values the step lengths need are read before any writer can have got there (a dual step) or
//     were left in the control block by k_chain_pre (a primal step); dx_p is computed by every
//     workgroup itself, with the owner's arithmetic; Binv0's rows are read BEFORE the barrier beta
//     is waited for (the eta file's share is added after it), so the column a deleting pivot moves
//     is no longer read by anybody.  The update therefore starts without waiting for anybody, and
//     the pivot's books are kept by ONE WAVE of workgroup 0 beside it.
//
//   What remains: the eta file's beta = W^T a_j must be complete before the eta share of FTRAN's
//   rows (1 barrier), a primal step's ratio test needs every row of dx (1 more).
//   Barriers per iteration: 2 (primal step: both in k_chain_pre), 1 (dual step: in k_chain_post).
//
// The barrier itself is fence-free (profiles/r02_gridsync_vs_kernel_boundary.txt): an agent-scope
// release would write back the XCD's whole L2.  What crosses a barrier -- 64 doubles of beta, one
// candidate record per workgroup -- is published with agent-scope (sc1, write-through) stores by
// lane 0 and read with sc1 loads; everything else a phase reads was written by an earlier launch or
// by the reading workgroup itself.  Arrivals are counted on eight counters (one per residue of the
// workgroup index mod 8, so that 256 atomics do not queue on one address), the last arrival of a
// residue class bumps the top counter everybody polls.  The counters only grow; the number of
// barriers passed so far lives in the control block (bar_gen).  A poll loop gives up after ~2.4 s (2^24 polls), marks
// the control block (status DZG_PANIC, bar_timeout) and the host returns DZG_E_DEVICE: every wave
// reaches its exit whatever happens (a barrier can only fail when the launch's workgroups are not
// all resident, i.e. when something else occupies CUs or LDS of this device).
//
// Arithmetic: the row, dot-product, book-keeping and update formulas are the functions the
// seven-launch kernels call (fast_rows.h), and argmax reductions do not depend on how candidates are
// grouped, so a solve is bit-identical whichever form runs an iteration; the host may switch between
// them at any poll (it does when the compact width outgrows the LDS copy of the gathered column).
#include <chrono>
#include "common.h"
#include "fast_decide.h"
#include "fast_rows.h"

#include "chain_barrier.h"

// Diagnostic (DZG_CHAIN_DEBUG=1): lane 0 of workgroup 0 accumulates the 100 MHz real-time clock
// between phase boundaries; dbg[16 * kernel_and_kind + stage] += ticks, dbg[.. + 15] += 1.
struct ChainStamps {
    unsigned long long *dbg;
    unsigned long long last;
    int stage;
    __device__ __forceinline__ void start(unsigned long long *p)
    {
        dbg = (blockIdx.x == 0 && threadIdx.x == 0) ? p : nullptr;
        stage = 0;
        if (dbg) last = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void mark(int slot)
    {
        if (!dbg) return;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        dbg[16 * slot + stage] += now - last;
        last = now;
        ++stage;
    }
    __device__ __forceinline__ void done(int slot)
    {
        if (dbg) dbg[16 * slot + 15] += 1;
    }
};

// ---------------------------------------------------------------------------------
// Work split.  Workgroup b owns rows [r0, r1) -- thread t the row r0 + t -- in FTRAN, BTRAN and the
// update, and columns [q0, q1) of z -- thread t the column q0 + t.  (The host runs the chain only
// when a workgroup's share is at most one row and one column per thread.)  A thread loads what it
// needs of its row / column as the first thing the kernel does, together with the control block:
// after a kernel boundary every first touch of a line costs a trip to memory (1-2 us), and these
// trips are taken side by side instead of one behind the other.
// ---------------------------------------------------------------------------------
#define CH_NW (CH_THREADS / 64)
#define CH_MAXP 16 // FTRAN passes (rows per lane group) whose partial sums wait in registers

__device__ __forceinline__ void chain_rows(int m, int &r0, int &r1)
{
    const int per = ((m + (int)gridDim.x - 1) / (int)gridDim.x + 3) & ~3;
    r0 = (int)blockIdx.x * per;
    if (r0 > m) r0 = m;
    r1 = r0 + per < m ? r0 + per : m;
}

__device__ __forceinline__ void chain_cols(int q, int &q0, int &q1)
{
    const int per = (q + (int)gridDim.x - 1) / (int)gridDim.x;
    q0 = (int)blockIdx.x * per;
    if (q0 > q) q0 = q;
    q1 = q0 + per < q ? q0 + per : q;
}

// up to 4 x 64 partial candidates in the registers of one wave, loaded before their count is known
struct ChainSpec {
    double r[4], h[4];
    int k[4];
};
__device__ __forceinline__ void chain_spec_load(ChainSpec &s, const double *__restrict__ pr,
                                                const int *__restrict__ pk,
                                                const double *__restrict__ ph, int lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { // (the arrays hold 4096 entries: reading past `count` is harmless)
        s.r[j] = pr[lane + 64 * j];
        s.k[j] = pk[lane + 64 * j];
        s.h[j] = ph[lane + 64 * j];
    }
}
__device__ __forceinline__ DzgCand2 chain_spec_reduce(const ChainSpec &s, int count, int lane)
{
    DzgCand2 best = dzg_cand2_none();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        DzgCand2 o;
        o.r = s.r[j];
        o.k = s.k[j];
        o.h = s.h[j];
        if (lane + 64 * j < count) best = dzg_better2(best, o);
    }
    return dzg_wave_best2(best);
}

// two argmax candidates through LDS: written by lane 0 of waves 0 and 1, read by everybody
struct ChainSlots {
    double r[2], h[2];
    int k[2];
};
__device__ __forceinline__ void chain_put(ChainSlots &s, int slot, const DzgCand2 &c)
{
    s.r[slot] = c.r;
    s.k[slot] = c.k;
    s.h[slot] = c.h;
}
__device__ __forceinline__ DzgCand2 chain_get(const ChainSlots &s, int slot)
{
    DzgCand2 c;
    c.r = s.r[slot];
    c.k = s.k[slot];
    c.h = s.h[slot];
    return c;
}

// best of the candidates threads 0 .. n-1 hold; valid in wave 0 (n <= 64: no workgroup barrier)
__device__ __forceinline__ DzgCand2 chain_best(DzgCand2 c, int n)
{
    if (n <= 64) return (threadIdx.x < 64) ? dzg_wave_best2(c) : c;
    return dzg_block_best2(c);
}

// The entering column gathered to compact coordinates, in LDS (zero-padded to an even length).
// dr0, dr1 = drow[tid], drow[tid + CH_THREADS], loaded early.
__device__ __forceinline__ void chain_stage_ag(double *s_ag, int k, int code,
                                               const double *__restrict__ a,
                                               const int *__restrict__ drow, int dr0, int dr1)
{
    const int k2 = (k + 1) & ~1, tid = threadIdx.x;
    const int rr = -1 - code; // (entering slack: the unit vector of its row)
    if (tid < k2) s_ag[tid] = tid < k ? (code < 0 ? (dr0 == rr ? 1.0 : 0.0) : a[dr0]) : 0.0;
    if (tid + CH_THREADS < k2)
        s_ag[tid + CH_THREADS] = tid + CH_THREADS < k ? (code < 0 ? (dr1 == rr ? 1.0 : 0.0) : a[dr1]) : 0.0;
    for (int c = tid + 2 * CH_THREADS; c < k2; c += CH_THREADS)
        s_ag[c] = c < k ? (code < 0 ? (drow[c] == rr ? 1.0 : 0.0) : a[drow[c]]) : 0.0;
}

// The entering column: the device's own matrix (one GPU, or a sharded rank that keeps every column),
// or -- a PARTITIONED rank, whose HBM holds its own block only -- the copy that travels in the
// winner's exchange record (8 header doubles, then the column; include/dantzig_amd.h).
template <bool SHARD>
__device__ __forceinline__ const double *chain_col(const DzgDev &d, int code,
                                                   const double *__restrict__ xrecv, int w_rec)
{
    if (code < 0) return nullptr;
    if (SHARD && d.xstride > 8) return xrecv + (long long)w_rec * d.xstride + 8;
    return d.A + (long long)(code - d.col0) * d.lda;
}

// beta_t = W_t . a_j by workgroup t, published for everybody
__device__ __forceinline__ void chain_beta(const DzgDev &d, int neta, int code,
                                           const double *__restrict__ a)
{
    // (block-uniform loop: a grid of fewer workgroups than pending etas -- a CU-masked or
    // partitioned device -- takes several rows per workgroup; same bits, beta_t depends on t only)
    for (int b = blockIdx.x; b < neta; b += gridDim.x) {
        const double *wt = d.W + (long long)b * d.ldw;
        double acc;
        if (code < 0)
            acc = wt[-1 - code];
        else
            acc = fast_beta_dot(wt, a, d.m);
        if (threadIdx.x == 0) st_sc1(d.beta + b, acc);
    }
}

// FTRAN on this workgroup's rows, first half (needs the gathered column only): lane group `grp` of
// wave `wave` takes row r0 + (pass * CH_NW + wave) * RPW + grp; the partial sums of the first
// CH_MAXP passes stay in registers across the barrier beta is waited for.
template <int LPR>
__device__ __forceinline__ void chain_dot_head(const DzgDev &d, int r0, int r1, int k,
                                               const double *s_ag, double (&accs)[CH_MAXP])
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, grp = lane / LPR;
    const int k2 = (k + 1) & ~1;
    // (block-uniform) how the rows are loaded: an inverse that outgrows the 256-MB Infinity Cache
    // streams past the caches (nontemporal loads, eight steps in flight: 125.4 -> 102.9 us per FTRAN
    // at k = 7 700 of config 3), a smaller one is re-read from it pivot after pivot and plain loads
    // are faster (55.4 against 60.4 us at k = 4 049): profiles/r04_ftran_row_loads_ab.txt
    const int variant = d.ftran_variant >= 0 ? d.ftran_variant : (k >= d.ftran_nt_k ? 2 : 0);
    if (LPR == 64 && variant == 1) {
#pragma unroll
        for (int pass = 0; pass < CH_MAXP; ++pass) {
            const int i = r0 + (pass * CH_NW + wave) * RPW + grp;
            accs[pass] = fast_gemv_row_head<LPR, 1>(i < r1 ? i : d.m, d.m, k2, d.binv, d.ldb, s_ag, sub);
        }
        return;
    }
    if (LPR == 64 && variant == 2) {
#pragma unroll
        for (int pass = 0; pass < CH_MAXP; ++pass) {
            const int i = r0 + (pass * CH_NW + wave) * RPW + grp;
            accs[pass] = fast_gemv_row_head<LPR, 2>(i < r1 ? i : d.m, d.m, k2, d.binv, d.ldb, s_ag, sub);
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < CH_MAXP; ++pass) {
        const int i = r0 + (pass * CH_NW + wave) * RPW + grp;
        accs[pass] = fast_gemv_row_head<LPR>(i < r1 ? i : d.m, d.m, k2, d.binv, d.ldb, s_ag, sub);
    }
}

// second half: the eta file's share and the sum over the lanes; row sums land in s_dx[row - r0]
template <int LPR>
__device__ __forceinline__ void chain_dot_tail(const DzgDev &d, int r0, int r1, int k, int neta,
                                               const double *s_ag, const double *s_beta,
                                               const double (&accs)[CH_MAXP], double *s_dx)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, grp = lane / LPR;
    const int k2 = (k + 1) & ~1;
    const int npass = (r1 - r0 + CH_NW * RPW - 1) / (CH_NW * RPW);
#pragma unroll
    for (int pass = 0; pass < CH_MAXP; ++pass) {
        if (pass < npass) { // (block-uniform)
            const int i = r0 + (pass * CH_NW + wave) * RPW + grp;
            const int ii = i < r1 ? i : d.m;
            const double acc = fast_gemv_row_tail<LPR>(accs[pass], ii, d.m, neta, d.U, d.ldw, s_beta, sub);
            if (ii < d.m && sub == 0) s_dx[i - r0] = acc;
        }
    }
    for (int pass = CH_MAXP; pass < npass; ++pass) { // very tall shares: the whole row now
        const int i = r0 + (pass * CH_NW + wave) * RPW + grp;
        const int ii = i < r1 ? i : d.m;
        const double acc = fast_gemv_row<LPR>(ii, d.m, k2, neta, d.binv, d.ldb, s_ag, d.U, d.ldw, s_beta, sub);
        if (ii < d.m && sub == 0) s_dx[i - r0] = acc;
    }
}

// ---------------------------------------------------------------------------------
// k_chain_pre: src/simplex.rs:274-306 (status), :226-229 + :439-461 (a primal step's FTRAN and
// ratio test), then v = row p of the inverse (:231-236's solve).  grid = one workgroup per CU.
// ---------------------------------------------------------------------------------
// SHARD (column sharding with the matrix replicated, one process per GPU): the z-side first pivot
// comes from the merge of every rank's proposal (xrecv: the first exchange's records) -- all ranks
// see the same records in the same order and apply the same rule, so they take the same decision.
template <bool SHARD>
__global__ __launch_bounds__(CH_THREADS) void k_chain_pre(const DzgDev d, unsigned long long *bar,
                                                          unsigned long long *dbg,
                                                          const double *__restrict__ xrecv)
{
    __shared__ double s_ag[CH_AGCAP];
    __shared__ double s_beta[R_], s_dx[CH_THREADS];
    __shared__ ChainSlots s_c;
    ChainStamps ts;
    ts.start(dbg);
    DzgCtl *ctl = d.ctl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = d.m, nwg = (int)gridDim.x;
    const bool lead = blockIdx.x == 0 && tid == 0;
    // ---- first touches, side by side: candidates of the last update, this thread's row
    ChainSpec sp;
    if (wave == 0 && !SHARD)
        chain_spec_load(sp, d.fpz_r, d.fpz_k, d.fpz_h, lane);
    else if (wave == 1)
        chain_spec_load(sp, d.fpx_r, d.fpx_k, d.fpx_h, lane);
    DzgCand2 cj_rec = dzg_cand2_none();
    int w_rec = -1;
    if (SHARD) w_rec = shard_merge(xrecv, d.xstride, d.world, cj_rec);
    int r0, r1;
    chain_rows(m, r0, r1);
    const int row = r0 + tid;
    const bool has_row = row < r1;
    int dslot_i = -1, bcode_i = 0;
    double x_i = 0.0, xbar_i = 0.0;
    if (has_row) {
        dslot_i = d.dslot[row];
        bcode_i = d.bcode[row];
        x_i = d.x[row];
        xbar_i = d.xbar[row];
    }
    DzgCtl c = *ctl; // one snapshot; nothing the lead lane writes below is read from it
    if (c.status != DZG_RUNNING) return;
    unsigned long long gen = c.bar_gen;
    const int neta = c.neta, k = c.ncompact;
    const int dr0 = tid < k ? d.drow[tid] : -1;
    const int dr1 = tid + CH_THREADS < k ? d.drow[tid + CH_THREADS] : -1;
    if (wave < 2 && !(SHARD && wave == 0)) {
        const DzgCand2 w = chain_spec_reduce(sp, c.fp_count, lane);
        if (lane == 0) chain_put(s_c, wave, w);
    }
    __syncthreads();
    const DzgCand2 cj = SHARD ? cj_rec : chain_get(s_c, 0), ci = chain_get(s_c, 1);
    int kind;
    double mu;
    if (!fast_status(ctl, c, lead, cj, ci, d.eps, m, SHARD, kind, &mu)) return;
    const int slot = kind == DZG_STEP_PRIMAL ? 0 : 1;
    ts.mark(slot); // 0: first touches + status
    if (k > CH_AGCAP || c.fp_count > 256) { // the host runs the seven launches before this can happen
        if (lead) ctl->status = DZG_PANIC;
        return;
    }
    if (lead) {
        ctl->neta_cur = neta;
        ctl->k_cur = k;
    }
    int p;
    if (kind == DZG_STEP_PRIMAL) {
        const int epos = cj.k;
        const int code = d.nbcode[epos];
        double zr = 0.0, zbr = 0.0;
        if (lead) {
            zr = d.z[epos];
            zbr = d.zbar[epos];
        }
        const double *a = chain_col<SHARD>(d, code, xrecv, w_rec);
        // this row's unit-column share of dx (fast_gemv_unit's term), loaded beside the gather
        const bool has_unit = has_row && bcode_i < 0;
        double unit_i = 0.0;
        if (has_unit) unit_i = code >= 0 ? a[-1 - bcode_i] : (code == bcode_i ? 1.0 : 0.0);
        int edslot = -1;
        if (lead && code < 0) edslot = d.dslot[-1 - code];
        chain_stage_ag(s_ag, k, code, a, d.drow, dr0, dr1);
        chain_beta(d, neta, code, a);
        if (lead) {
            ctl->enter_code = code;
            ctl->zr = zr; // (SHARD: z is kept by the column's owner; k_chain_post takes it from
            ctl->zbar_r = zbr; //  the second exchange's records instead)
            ctl->enter_dslot = edslot;
            if (SHARD) ctl->enter_src = w_rec;
        }
        __syncthreads(); // the gathered column is complete
        double accs[CH_MAXP];
        if (k > 512)
            chain_dot_head<64>(d, r0, r1, k, s_ag, accs);
        else
            chain_dot_head<16>(d, r0, r1, k, s_ag, accs);
        ts.mark(slot); // 1: beta, gather, Binv0 rows
        if (!chain_barrier(ctl, bar, gen)) return;
        ts.mark(slot); // 2: barrier
        if (tid < R_) s_beta[tid] = tid < neta ? ld_sc1(d.beta + tid) : 0.0;
        __syncthreads();
        if (k > 512)
            chain_dot_tail<64>(d, r0, r1, k, neta, s_ag, s_beta, accs, s_dx);
        else
            chain_dot_tail<16>(d, r0, r1, k, neta, s_ag, s_beta, accs, s_dx);
        __syncthreads();
        // ratio test on this thread's row (src/simplex.rs:439-461)
        DzgCand2 best = dzg_cand2_none();
        if (has_row) {
            double dxi = s_dx[tid];
            if (has_unit) dxi += unit_i;
            d.dx[row] = dxi;
            const double scaled = mu * xbar_i;
            const double den = x_i + scaled;
            DzgCand2 cnd;
            cnd.r = dzg_div(dxi, den);
            cnd.k = row;
            cnd.h = -__builtin_inf();
            if (cnd.r > 0.0) best = dzg_better2(best, cnd);
            if (dzg_noise_zero(den, x_i, scaled, c.tau)) best.h = __builtin_inf();
        }
        best = chain_best(best, r1 - r0);
        if (tid == 0) {
            st_sc1(d.rx_r + blockIdx.x, best.r);
            st_sc1(d.rx_k + blockIdx.x, best.k);
            st_sc1(d.rx_h + blockIdx.x, best.h);
        }
        ts.mark(slot); // 3: eta share, candidates
        if (!chain_barrier(ctl, bar, gen)) return;
        ts.mark(slot); // 4: barrier
        if (wave == 0) {
            DzgCand2 w = dzg_cand2_none();
            for (int i = lane; i < nwg; i += 64) {
                DzgCand2 o;
                o.r = ld_sc1(d.rx_r + i);
                o.k = ld_sc1(d.rx_k + i);
                o.h = ld_sc1(d.rx_h + i);
                w = dzg_better2(w, o);
            }
            w = dzg_wave_best2(w);
            if (lane == 0) chain_put(s_c, 0, w);
        }
        __syncthreads();
        const DzgCand2 cw = chain_get(s_c, 0);
        if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_UNBOUNDED)) { // :313
            if (lead) ctl->bar_gen = gen;
            return;
        }
        p = cw.k;
        if (lead) ctl->leave_pos = p;
        ts.mark(slot); // 5: ratio test
    } else {
        p = ci.k;
    }
    // ---- BTRAN: v = row p of Binv on this thread's row; one trip: nothing below waits for another load
    const int lcode = d.bcode[p];
    if (has_row) {
        const double base = dslot_i >= 0 ? d.binv[(long long)p * d.ldb + dslot_i] : (lcode == -1 - row ? 1.0 : 0.0);
        const double vr = base - fast_btran_eta(neta, d.U, d.ldw, d.W, d.ldw, p, row);
        d.v[row] = vr;
        if (dslot_i >= 0 && d.vc) d.vc[dslot_i] = vr; // (compact copy: the row-wise pricing pass's coefficients)
    }
    if (lead) {
        ctl->xp = d.x[p];
        ctl->xbp = d.xbar[p];
        ctl->leave_code = lcode;
        if (gen != c.bar_gen) ctl->bar_gen = gen;
    }
    ts.mark(slot); // primal 6 / dual 1: BTRAN row
    ts.done(slot);
}

// dz of ONE nonbasic position as the finishing launch of the row-wise pricing pass computes it
// (k_price_rows_finish, k_price_kernels.h): the G partial sums of the row groups in group order for
// a structural column, the unit column's single product otherwise.
__device__ __forceinline__ double chain_fold_dz(const DzgDev &d, int code, int G)
{
    if (code < 0) {
        const double p = 1.0 * -d.v[-1 - code];
        return 0.0 + p; // Iterator::sum identity + the single stored entry
    }
    const double *src = d.ppart + (code - d.col0);
    double sum = 0.0;
    int gg = 0;
    for (; gg + 8 <= G; gg += 8) { // eight partials side by side, added in group order
        double t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = src[(long long)(gg + e) * d.ldt];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum = sum + t[e];
    }
    for (; gg < G; ++gg) sum = sum + src[(long long)gg * d.ldt];
    return -sum;
}

// ---------------------------------------------------------------------------------
// k_chain_post: a dual step's ratio test and FTRAN (src/simplex.rs:324-325, :226-229), the pivot's
// books (fast_rows.h; the last wave of workgroup 0, beside everybody's update), pivot() x4
// (:262-265, :410-421) with the eta append, and the first-pivot candidates of the next iteration
// (:423-437).  only_partials != 0: the candidates only.
// ---------------------------------------------------------------------------------
// SHARD: the ratio test of a dual step is the merge of the ranks' proposals (xrecv: the second
// exchange's records), z, zbar, dz of the entering position come from its owner's record, and z is
// only kept -- and offered as a first-pivot candidate -- for slack positions and owned columns.
// FOLD (one GPU, the row-wise pricing pass, round 4): the pass's FINISHING launch is this kernel's
// head -- a workgroup owns its columns of z anyway: thread t adds the row groups' partial sums of
// position q0 + t itself (the finishing launch's arithmetic, chain_fold_dz), a dual step's ratio
// candidates are reduced per workgroup and cross one more device-wide barrier (a primal step needs
// none), and dz_r of the entering position is re-derived by every workgroup.  One launch and its
// cold first touches less per pivot; the same sums and candidates, bit for bit.
template <bool SHARD>
__global__ __launch_bounds__(CH_THREADS) void k_chain_post(const DzgDev d, unsigned long long *bar,
                                                           const DzgPivotArgs pa, int only_partials,
                                                           int nrz, unsigned long long *dbg,
                                                           const double *__restrict__ xrecv, int fold)
{
    __shared__ double s_ag[CH_AGCAP];
    __shared__ double s_beta[R_], s_dx[CH_THREADS];
    __shared__ double s_dxp;
    __shared__ ChainSlots s_c;
    ChainStamps ts;
    ts.start(only_partials ? nullptr : dbg);
    int slot = 2;
    DzgCtl *ctl = d.ctl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = d.m, q = d.q;
    const bool lead = blockIdx.x == 0 && tid == 0;
    // ---- first touches, side by side: the pricing pass's candidates, this thread's row and column
    DzgCand2 mine = dzg_cand2_none();
    ChainSpec sp;
    const bool one_wave = nrz <= 256; // (block-uniform) the candidates fit one wave's registers
    if (SHARD || fold) {
        // (the candidates travel in the exchange records / are formed below)
    } else if (!only_partials && one_wave) {
        if (wave == 0) chain_spec_load(sp, d.rz_r, d.rz_k, d.rz_h, lane);
    } else if (!only_partials) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { // (4096 entries: harmless past nrz; a primal step ignores them)
            const int i = tid + CH_THREADS * j;
            DzgCand2 o;
            o.r = d.rz_r[i];
            o.k = d.rz_k[i];
            o.h = d.rz_h[i];
            if (i < nrz) mine = dzg_better2(mine, o);
        }
        for (int i = tid + 4 * CH_THREADS; i < nrz; i += CH_THREADS) {
            DzgCand2 o;
            o.r = d.rz_r[i];
            o.k = d.rz_k[i];
            o.h = d.rz_h[i];
            mine = dzg_better2(mine, o);
        }
    }
    int r0, r1, q0, q1;
    chain_rows(m, r0, r1);
    chain_cols(q, q0, q1);
    const int row = r0 + tid, col = q0 + tid;
    const bool has_row = row < r1, has_col = col < q1;
    double x_i = 0.0, xbar_i = 0.0, v_i = 0.0, dx_i = 0.0, z_k = 0.0, zbar_k = 0.0, dz_k = 0.0;
    int bcode_i = 0, nbcode_k = -1;
    if (has_row) {
        x_i = d.x[row];
        xbar_i = d.xbar[row];
        if (!only_partials) {
            v_i = d.v[row];
            dx_i = d.dx[row]; // (a dual step computes its own below)
            bcode_i = d.bcode[row];
        }
    }
    if (has_col) {
        z_k = d.z[col];
        zbar_k = d.zbar[col];
        if (!only_partials && !fold) dz_k = d.dz[col];
        if (SHARD || fold) nbcode_k = d.nbcode[col]; // (before the books of this launch rewrite position r's)
    }
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    unsigned long long gen = c.bar_gen;
    // what the update needs (all block-uniform)
    int p = 0, r = 0, teta = 0, wzero = -1, del_ce = -1, del_last = -1, app_col = -1, new_code_r = -1;
    double t = 0.0, s = 0.0, tbar = 0.0, sbar = 0.0, rdxp = 0.0, tau = c.tau;
    if (!only_partials) {
        const int neta = c.neta_cur, k = c.k_cur, kind = c.kind;
        const int ci = c.leave_code;
        const double xp = c.xp, xbp = c.xbp;
        p = c.leave_pos;
        int cj, edslot;
        double zr, zbr, dzr, dxp;
        int G = (k + 1 + DZG_PR_BATCH - 1) / DZG_PR_BATCH; // row groups of the pricing pass (FOLD)
        G = G > DZG_PR_GMAX ? DZG_PR_GMAX : G;
        if (!SHARD && fold) {
            // this workgroup's (position, row group) partial sums, fetched by ALL its threads in one
            // trip through LDS (the gathered column's buffer is free until FTRAN); the owner of a
            // position then adds its G values in group order -- chain_fold_dz's sum, without 32
            // loads queuing behind one thread
            const int ncols = q1 - q0;
            int *s_code = reinterpret_cast<int *>(s_dx); // (CH_THREADS doubles: room for 512 codes)
            if (has_col) s_code[tid] = nbcode_k;
            __syncthreads();
            if ((long long)ncols * G <= CH_AGCAP) {
                for (int idx = tid; idx < ncols * G; idx += CH_THREADS) {
                    const int g = idx / ncols, pl = idx - g * ncols;
                    const int code = s_code[pl];
                    s_ag[idx] = code >= 0 ? d.ppart[(long long)g * d.ldt + (code - d.col0)] : 0.0;
                }
                __syncthreads();
                if (has_col) {
                    if (nbcode_k < 0) {
                        dz_k = chain_fold_dz(d, nbcode_k, G);
                    } else {
                        double sum = 0.0;
                        for (int g = 0; g < G; ++g) sum = sum + s_ag[g * ncols + tid];
                        dz_k = -sum;
                    }
                }
                __syncthreads(); // (s_ag and s_dx are about to be reused)
            } else if (has_col) {
                dz_k = chain_fold_dz(d, nbcode_k, G);
            }
        }
        if (kind == DZG_STEP_DUAL) {
            const int dr0 = tid < k ? d.drow[tid] : -1;
            const int dr1 = tid + CH_THREADS < k ? d.drow[tid + CH_THREADS] : -1;
            DzgCand2 cw;
            int w_rec = -1;
            if (SHARD) {
                w_rec = shard_merge(xrecv, d.xstride, d.world, cw);
            } else if (fold) {
                // the ratio test of the z side (src/simplex.rs:449-460) on this workgroup's columns;
                // the workgroups' winners cross a barrier, everybody reduces them
                DzgCand2 cnd = dzg_cand2_none();
                if (has_col) {
                    const double scaled = c.mu * zbar_k, den = z_k + scaled;
                    DzgCand2 o;
                    o.r = dzg_div(dz_k, den);
                    o.k = col;
                    o.h = -__builtin_inf();
                    if (o.r > 0.0) cnd = dzg_better2(cnd, o);
                    if (dzg_noise_zero(den, z_k, scaled, c.tau)) cnd.h = __builtin_inf();
                }
                cnd = chain_best(cnd, q1 - q0);
                if (tid == 0) {
                    st_sc1(d.rz_r + blockIdx.x, cnd.r);
                    st_sc1(d.rz_k + blockIdx.x, cnd.k);
                    st_sc1(d.rz_h + blockIdx.x, cnd.h);
                }
                if (!chain_barrier(ctl, bar, gen)) return;
                if (wave == 0) {
                    DzgCand2 w = dzg_cand2_none();
                    for (int i = lane; i < (int)gridDim.x; i += 64) {
                        DzgCand2 o;
                        o.r = ld_sc1(d.rz_r + i);
                        o.k = ld_sc1(d.rz_k + i);
                        o.h = ld_sc1(d.rz_h + i);
                        w = dzg_better2(w, o);
                    }
                    w = dzg_wave_best2(w);
                    if (lane == 0) chain_put(s_c, 0, w);
                }
                __syncthreads();
                cw = chain_get(s_c, 0);
            } else if (one_wave) {
                if (wave == 0) {
                    const DzgCand2 w = chain_spec_reduce(sp, nrz, lane);
                    if (lane == 0) chain_put(s_c, 0, w);
                }
                __syncthreads();
                cw = chain_get(s_c, 0);
            } else {
                cw = dzg_block_best2(mine);
            }
            if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) { // :325
                if (lead && gen != c.bar_gen) ctl->bar_gen = gen;
                return;
            }
            slot = 3;
            ts.mark(slot); // 0: first touches + ratio test
            r = cw.k;
            if (SHARD) { // the entering column's code and z, zbar, dz travel in the winner's record
                const double *rec = xrecv + (long long)w_rec * d.xstride;
                cj = (int)rec[5];
                zr = rec[2];
                zbr = rec[3];
                dzr = rec[4];
                if (lead) {
                    ctl->enter_src = w_rec;
                    ctl->zr = zr;
                    ctl->zbar_r = zbr;
                    ctl->dz_r = dzr;
                    ctl->use_record = 1;
                }
            } else {
                cj = d.nbcode[r];
                zr = d.z[r];
                zbr = d.zbar[r];
                dzr = fold ? chain_fold_dz(d, cj, G) : d.dz[r];
            }
            const double *a = chain_col<SHARD>(d, cj, xrecv, w_rec);
            edslot = cj < 0 ? d.dslot[-1 - cj] : -1;
            const bool has_unit = has_row && bcode_i < 0;
            double unit_i = 0.0, unit_p = 0.0;
            if (has_unit) unit_i = cj >= 0 ? a[-1 - bcode_i] : (cj == bcode_i ? 1.0 : 0.0);
            if (ci < 0) unit_p = cj >= 0 ? a[-1 - ci] : (cj == ci ? 1.0 : 0.0);
            chain_stage_ag(s_ag, k, cj, a, d.drow, dr0, dr1);
            chain_beta(d, neta, cj, a);
            if (lead) {
                ctl->enter_pos = r;
                ctl->enter_code = cj;
            }
            c.enter_pos = r;
            c.enter_code = cj;
            __syncthreads(); // the gathered column is complete
            // Binv0's share of this workgroup's rows, and of row p: every workgroup computes dx_p
            // itself (the arithmetic of the row's owner, so the same bits) and then needs nobody
            // else's result for the step lengths
            const int k2 = (k + 1) & ~1;
            double accs[CH_MAXP], accp;
            if (k > 512) {
                chain_dot_head<64>(d, r0, r1, k, s_ag, accs);
                accp = fast_gemv_row_head<64>(wave == CH_NW - 1 ? p : m, m, k2, d.binv, d.ldb, s_ag, lane);
            } else {
                chain_dot_head<16>(d, r0, r1, k, s_ag, accs);
                accp = fast_gemv_row_head<16>(wave == CH_NW - 1 && lane < 16 ? p : m, m, k2, d.binv, d.ldb, s_ag,
                                              lane % 16);
            }
            ts.mark(slot); // 1: loads, beta, gather, Binv0 rows
            if (!chain_barrier(ctl, bar, gen)) return;
            ts.mark(slot); // 2: barrier
            if (tid < R_) s_beta[tid] = tid < neta ? ld_sc1(d.beta + tid) : 0.0;
            __syncthreads();
            if (k > 512) {
                chain_dot_tail<64>(d, r0, r1, k, neta, s_ag, s_beta, accs, s_dx);
                if (wave == CH_NW - 1) {
                    const double acc = fast_gemv_row_tail<64>(accp, p, m, neta, d.U, d.ldw, s_beta, lane);
                    if (lane == 0) s_dxp = ci < 0 ? acc + unit_p : acc;
                }
            } else {
                chain_dot_tail<16>(d, r0, r1, k, neta, s_ag, s_beta, accs, s_dx);
                if (wave == CH_NW - 1) {
                    const double acc = fast_gemv_row_tail<16>(accp, lane < 16 ? p : m, m, neta, d.U, d.ldw,
                                                              s_beta, lane % 16);
                    if (lane == 0) s_dxp = ci < 0 ? acc + unit_p : acc;
                }
            }
            __syncthreads();
            dxp = s_dxp;
            if (has_row) {
                dx_i = s_dx[tid];
                if (has_unit) dx_i += unit_i;
                d.dx[row] = dx_i;
            }
            ts.mark(slot); // 3: eta share
        } else {
            r = c.enter_pos;
            cj = c.enter_code;
            zr = c.zr;
            zbr = c.zbar_r;
            dzr = (!SHARD && fold) ? chain_fold_dz(d, cj, G) : d.dz[r];
            if (SHARD) { // the owner of the entering position has published z, zbar, dz
                int w_rec = -1;
                for (int rk = 0; rk < d.world && w_rec < 0; ++rk)
                    if ((int)xrecv[(long long)rk * d.xstride + 1] == r) w_rec = rk;
                if (w_rec < 0) { // no rank owns the entering position: cannot happen
                    if (lead) ctl->status = DZG_PANIC;
                    return;
                }
                const double *rec = xrecv + (long long)w_rec * d.xstride;
                zr = rec[2];
                zbr = rec[3];
                dzr = rec[4];
                if (lead) {
                    ctl->zr = zr;
                    ctl->zbar_r = zbr;
                    ctl->dz_r = dzr;
                    ctl->use_record = 1;
                }
            }
            edslot = c.enter_dslot;
            dxp = d.dx[p];
            ts.mark(slot); // 0: first touches
        }
        const DzgPivotScalars ps = fast_pivot_scalars(xp, xbp, dxp, zr, zbr, dzr, neta, c.max_pivot_err);
        const bool appended = ci < 0, deleted = cj < 0;
        new_code_r = ci; // the leaving variable takes nonbasic position r
        if (blockIdx.x == 0 && wave == CH_NW - 1) { // one wave keeps the books
            c.neta = neta;
            c.ncompact = k;
            fast_pivot_books_s(ctl, c, pa, ps, 1);
        }
        if (!ps.ok) { // (the books have set DZG_PANIC, src/simplex.rs:466)
            if (lead && gen != c.bar_gen) ctl->bar_gen = gen;
            return;
        }
        if (deleted) {
            del_last = k + (appended ? 1 : 0) - 1;
            del_ce = edslot;
        }
        if (appended) app_col = k;
        if (lead && gen != c.bar_gen) ctl->bar_gen = gen;
        t = ps.t;
        s = ps.s;
        tbar = ps.tbar;
        sbar = ps.sbar;
        teta = neta; // index of the eta appended now
        wzero = cj < 0 ? -1 - cj : -1;
        rdxp = 1.0 / dxp;
        if (c.tie_tol >= 0.0) { // the tolerance the books are setting
            double adaptive = 64.0 * ps.max_err;
            if (c.drift_tau > adaptive) adaptive = c.drift_tau;
            tau = adaptive > c.tie_tol ? adaptive : c.tie_tol;
        }
        ts.mark(slot); // primal 1 / dual 4: step lengths
    }
    // ---- update of this thread's row and column; candidates on the updated values
    DzgCand2 bx = dzg_cand2_none(), bz = dzg_cand2_none();
    if (has_row) {
        double xi = x_i, xb = xbar_i;
        if (!only_partials) {
            const double a = t * dx_i, b = tbar * dx_i;
            xi = (row == p) ? t : xi - a;
            xb = (row == p) ? tbar : xb - b;
            d.x[row] = xi;
            d.xbar[row] = xb;
            d.U[(long long)teta * d.ldw + row] = (row == p ? dx_i - 1.0 : dx_i) * rdxp;
            d.W[(long long)teta * d.ldw + row] = (row == wzero) ? 0.0 : v_i;
            // Binv0's columns: a leaving slack appends e_p, an entering slack deletes its column
            // (the last one takes its place).  Nobody reads Binv0 any more in this launch.
            double *brow = d.binv + (long long)row * d.ldb;
            if (del_last >= 0) {
                const double last_val = (app_col == del_last) ? (row == p ? 1.0 : 0.0) : brow[del_last];
                if (del_ce != del_last) brow[del_ce] = last_val;
                brow[del_last] = 0.0;
            } else if (app_col >= 0 && row == p) {
                brow[app_col] = 1.0;
            }
        }
        dzg_first_pivot_entry(bx, xi, xb, row, tau);
    }
    if (has_col) {
        double zk = z_k, zb = zbar_k;
        if (!only_partials) {
            const double a = s * dz_k, b = sbar * dz_k;
            zk = (col == r) ? s : zk - a;
            zb = (col == r) ? sbar : zb - b;
            d.z[col] = zk;
            d.zbar[col] = zb;
        }
        bool mine = true; // SHARD: z is only kept for slack positions and owned columns
        if (SHARD) {
            const int code = (!only_partials && col == r) ? new_code_r : nbcode_k;
            mine = code < 0 || (code >= d.col0 && code < d.col1);
        }
        if (mine) dzg_first_pivot_entry(bz, zk, zb, col, tau);
    }
    bx = chain_best(bx, r1 - r0);
    bz = chain_best(bz, q1 - q0);
    if (tid == 0) {
        d.fpx_r[blockIdx.x] = bx.r;
        d.fpx_k[blockIdx.x] = bx.k;
        d.fpx_h[blockIdx.x] = bx.h;
        d.fpz_r[blockIdx.x] = bz.r;
        d.fpz_k[blockIdx.x] = bz.k;
        d.fpz_h[blockIdx.x] = bz.h;
        if (blockIdx.x == 0) ctl->fp_count = (int)gridDim.x;
    }
    ts.mark(slot); // primal 2 / dual 5: update + candidates
    ts.done(slot);
}

static_assert(CH_BAR_WORDS == DZG_CHAIN_BAR_WORDS, "barrier counter block");

// Residency of the chain kernels as the runtime computes it (registers, LDS, 512 threads): the
// smallest count over the four instantiations; 0 = some kernel does not fit a CU at all.
int dzg_chain_resident_per_cu(void)
{
    int least = 1 << 20, n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_chain_pre<false>, CH_THREADS, 0) != hipSuccess) return 0;
    least = n < least ? n : least;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_chain_pre<true>, CH_THREADS, 0) != hipSuccess) return 0;
    least = n < least ? n : least;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_chain_post<false>, CH_THREADS, 0) != hipSuccess) return 0;
    least = n < least ? n : least;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_chain_post<true>, CH_THREADS, 0) != hipSuccess) return 0;
    least = n < least ? n : least;
    return least;
}

// xrecv != nullptr: a column-sharded rank (replicated matrix), records of the exchange just done
void dzg_launch_chain_pre(const DzgDev &d, int grid, unsigned long long *bar,
                          unsigned long long *dbg, const double *xrecv, hipStream_t st)
{
    if (xrecv)
        hipLaunchKernelGGL(k_chain_pre<true>, dim3(grid), dim3(CH_THREADS), 0, st, d, bar, dbg, xrecv);
    else
        hipLaunchKernelGGL(k_chain_pre<false>, dim3(grid), dim3(CH_THREADS), 0, st, d, bar, dbg, xrecv);
}

void dzg_launch_chain_post(const DzgDev &d, int grid, unsigned long long *bar,
                           unsigned long long *dbg, int only_partials, int nrz, const double *xrecv,
                           hipStream_t st, int fold, int price_small)
{
    DzgPivotArgs pa = dzg_pivot_args(d);
    pa.price_small = price_small;
    if (xrecv)
        hipLaunchKernelGGL(k_chain_post<true>, dim3(grid), dim3(CH_THREADS), 0, st, d, bar,
                           pa, only_partials, nrz, dbg, xrecv, 0);
    else
        hipLaunchKernelGGL(k_chain_post<false>, dim3(grid), dim3(CH_THREADS), 0, st, d, bar,
                           pa, only_partials, nrz, dbg, xrecv, fold);
}

// ---------------------------------------------------------------------------------
// Test hook: a co-tenant on the device.  Each workgroup keeps 128 KB of LDS (so that no chain
// workgroup -- 136 KB -- fits beside it on the CU) and watches the 100 MHz clock until the time is
// up: an exit every wave reaches.  Launched on a stream of its own.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_hold(unsigned long long ticks, int *sink, int *started)
{
    extern __shared__ double s_hold[];
    s_hold[threadIdx.x] = 1.0;
    // (host-visible: the host waits until every workgroup of the co-tenant holds its CU)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long spins = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && spins < (1ull << 34)) {
        __builtin_amdgcn_s_sleep(64);
        ++spins;
    }
    if (s_hold[threadIdx.x] == 2.0) *sink = 1; // (keeps the allocation alive)
}

static hipStream_t g_hold_stream = nullptr;
static int *g_hold_sink = nullptr;
static int *g_hold_started = nullptr; // pinned host memory

// Returns once every workgroup of the co-tenant is resident (or DZG_E_DEVICE after two seconds):
// whether it gets in the way must not depend on how the two streams' first launches race.
extern "C" int dzg_debug_hold_cus(int32_t device, int32_t workgroups, double seconds)
{
    if (workgroups < 1 || workgroups > 256 || !(seconds > 0.0) || seconds > 30.0) return DZG_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return DZG_E_DEVICE;
    const int lds = 128 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_debug_hold),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return DZG_E_DEVICE;
    if (!g_hold_stream && hipStreamCreateWithFlags(&g_hold_stream, hipStreamNonBlocking) != hipSuccess)
        return DZG_E_DEVICE;
    if (!g_hold_sink && hipMalloc(&g_hold_sink, sizeof(int)) != hipSuccess) return DZG_E_NOMEM;
    if (!g_hold_started && hipHostMalloc(&g_hold_started, sizeof(int), hipHostMallocCoherent) != hipSuccess)
        return DZG_E_NOMEM;
    *(volatile int *)g_hold_started = 0;
    hipLaunchKernelGGL(k_debug_hold, dim3(workgroups), dim3(64), lds, g_hold_stream,
                       (unsigned long long)(seconds * 1e8), g_hold_sink, g_hold_started);
    if (hipGetLastError() != hipSuccess) return DZG_E_DEVICE;
    const auto t0 = std::chrono::steady_clock::now();
    while (*(volatile int *)g_hold_started < workgroups) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) return DZG_E_DEVICE;
    }
    return 0;
}

extern "C" int dzg_debug_hold_wait(void)
{
    if (!g_hold_stream) return 0;
    return hipStreamSynchronize(g_hold_stream) == hipSuccess ? 0 : DZG_E_DEVICE;
}

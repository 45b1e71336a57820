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
//   * A workgroup owns the same rows in FTRAN and in the update, and the same columns of z; the six
//     values the step lengths need are read before any workgroup can have rewritten them (a dual
//     step), or were left in the control block by k_chain_pre (a primal step).  The update
//     therefore starts without waiting for anybody.
//
//   What remains: the eta file's beta = W^T a_j must be complete before FTRAN's rows (1 barrier), a
//   primal step's ratio test needs every row of dx (1 more), a compact column that is deleted must
//   not be read any more (1, entering slacks only).  Barriers per iteration: 2 (primal), 1 (dual).
//
// The barrier itself is fence-free (profiles/r02_gridsync_vs_kernel_boundary.txt): an agent-scope
// release would write back the XCD's whole L2.  What crosses a barrier -- 64 doubles of beta, one
// candidate record per workgroup -- is published with agent-scope (sc1, write-through) stores by
// lane 0 and read with sc1 loads; everything else a phase reads was written by an earlier launch or
// by the reading workgroup itself.  Arrivals are counted on eight counters (one per residue of the
// workgroup index mod 8, so that 256 atomics do not queue on one address), the last arrival of a
// residue class bumps the top counter everybody polls.  The counters only grow; the number of
// barriers passed so far lives in the control block (bar_gen).  A poll loop gives up after ~1 s and
// ends the solve with DZG_PANIC: every wave reaches its exit whatever happens.
//
// Arithmetic: the row, dot-product, book-keeping and update formulas are the functions the
// seven-launch kernels call (fast_rows.h), and argmax reductions do not depend on how candidates are
// grouped, so a solve is bit-identical whichever form runs an iteration; the host may switch between
// them at any poll (it does when the compact width outgrows the LDS copy of the gathered column).
#include "common.h"
#include "fast_decide.h"
#include "fast_rows.h"

#define CH_THREADS 512
#define CH_AGCAP DZG_CHAIN_AGCAP
#define CH_GROUPS 8
#define CH_PAD 16 // counters 128 bytes apart

__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_sc1(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Device-wide barrier.  Returns false when it timed out (the caller returns at once).
__device__ __forceinline__ bool chain_barrier(DzgCtl *ctl, unsigned long long *bar,
                                              unsigned long long &gen)
{
    __shared__ int s_bar_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned grp = blockIdx.x % CH_GROUPS;
        const unsigned long long members = (gridDim.x - grp + CH_GROUPS - 1) / CH_GROUPS;
        const unsigned long long ngroups = gridDim.x < CH_GROUPS ? gridDim.x : CH_GROUPS;
        __builtin_amdgcn_s_waitcnt(0); // this lane's sc1 stores have left the CU
        const unsigned long long old = __hip_atomic_fetch_add(
            bar + (size_t)CH_PAD * (1 + grp), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * members)
            __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = gen * ngroups;
        int ok = 0;
        for (int spin = 0; spin < (1 << 21); ++spin) {
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) ctl->status = DZG_PANIC; // a workgroup never arrived: give up, loudly
        s_bar_ok = ok;
    }
    __syncthreads();
    return s_bar_ok != 0;
}

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

// rows [r0, r1) of workgroup b: the same split in FTRAN, BTRAN and the update
__device__ __forceinline__ void chain_rows(int m, int &r0, int &r1)
{
    const int per = ((m + (int)gridDim.x - 1) / (int)gridDim.x + 3) & ~3;
    r0 = (int)blockIdx.x * per;
    r1 = r0 + per < m ? r0 + per : m;
    if (r0 > m) r0 = m;
}

// The entering column gathered to compact coordinates, in LDS (zero-padded to an even length).
__device__ __forceinline__ void chain_stage_ag(double *s_ag, int k, int code,
                                               const double *__restrict__ a,
                                               const int *__restrict__ drow)
{
    const int k2 = (k + 1) & ~1;
    if (code < 0) {
        const int rr = -1 - code;
        for (int c = threadIdx.x; c < k2; c += blockDim.x) s_ag[c] = (c < k && drow[c] == rr) ? 1.0 : 0.0;
    } else {
        for (int c = threadIdx.x; c < k2; c += blockDim.x) s_ag[c] = c < k ? a[drow[c]] : 0.0;
    }
}

// beta_t = W_t . a_j by workgroup t, published for everybody
__device__ __forceinline__ void chain_beta(const DzgDev &d, int neta, int code,
                                           const double *__restrict__ a)
{
    const int b = blockIdx.x;
    if (b >= neta) return; // (block-uniform)
    const double *wt = d.W + (long long)b * d.ldw;
    double acc;
    if (code < 0)
        acc = wt[-1 - code];
    else
        acc = fast_beta_dot(wt, a, d.m);
    if (threadIdx.x == 0) st_sc1(d.beta + b, acc);
}

// dx on this workgroup's rows (+ a primal step's ratio candidates, src/simplex.rs:439-461)
template <int LPR>
__device__ __forceinline__ void chain_gemv(const DzgDev &d, int kind, int k, int neta, int code,
                                           const double *__restrict__ acolp, const double *s_ag,
                                           const double *s_beta, double mu, double tau,
                                           DzgCand2 &best)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int sub = lane % LPR, grp = lane / LPR;
    const int k2 = (k + 1) & ~1;
    int r0, r1;
    chain_rows(d.m, r0, r1);
    for (int i0 = r0 + wave * RPW; i0 < r1; i0 += nw * RPW) {
        const int i = i0 + grp;
        const int ii = i < r1 ? i : d.m; // rows beyond the chunk belong to the next workgroup
        double acc = fast_gemv_row<LPR>(ii, d.m, k2, neta, d.binv, d.ldb, s_ag, d.U, d.ldw, s_beta, sub);
        if (ii < d.m && sub == 0) {
            acc = fast_gemv_unit(acc, d.var_col[d.basis[i]], code, acolp);
            d.dx[i] = acc;
            if (kind == DZG_STEP_PRIMAL) {
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

// ---------------------------------------------------------------------------------
// k_chain_pre: src/simplex.rs:274-306 (status), :226-229 + :439-461 (a primal step's FTRAN and
// ratio test), then v = row p of the inverse (:231-236's solve).  grid = one workgroup per CU.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(CH_THREADS) void k_chain_pre(const DzgDev d, unsigned long long *bar,
                                                          unsigned long long *dbg)
{
    ChainStamps ts;
    ts.start(dbg);
    __shared__ double s_ag[CH_AGCAP];
    __shared__ double s_beta[R_], s_up[R_];
    DzgCtl *ctl = d.ctl;
    DzgCtl c = *ctl; // one snapshot; nothing the lead lane writes below is read from it
    if (c.status != DZG_RUNNING) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int m = d.m, nwg = (int)gridDim.x;
    unsigned long long gen = c.bar_gen;
    const int neta = c.neta, k = c.ncompact;
    const DzgCand2 cj = reduce_partials(d.fpz_r, d.fpz_k, d.fpz_h, c.fp_count);
    const DzgCand2 ci = reduce_partials(d.fpx_r, d.fpx_k, d.fpx_h, c.fp_count);
    int kind;
    double mu;
    if (!fast_status(ctl, c, lead, cj, ci, d.eps, m, false, kind, &mu)) return;
    const int slot = kind == DZG_STEP_PRIMAL ? 0 : 1;
    ts.mark(slot); // 0: snapshot + status
    if (k > CH_AGCAP) { // the host switches to the seven launches before this can happen
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
        const int code = d.var_col[d.nonbasis[epos]];
        if (lead) {
            ctl->enter_code = code;
            ctl->zr = d.z[epos];
            ctl->zbar_r = d.zbar[epos];
            ctl->enter_dslot = code < 0 ? d.dslot[-1 - code] : -1;
        }
        const double *a = code < 0 ? nullptr : d.A + (long long)(code - d.col0) * d.lda;
        chain_beta(d, neta, code, a);
        chain_stage_ag(s_ag, k, code, a, d.drow);
        ts.mark(slot); // 1: beta + gather
        if (!chain_barrier(ctl, bar, gen)) return;
        ts.mark(slot); // 2: barrier
        if (threadIdx.x < R_) s_beta[threadIdx.x] = (int)threadIdx.x < neta ? ld_sc1(d.beta + threadIdx.x) : 0.0;
        __syncthreads();
        DzgCand2 best = dzg_cand2_none();
        if (k > 512)
            chain_gemv<64>(d, kind, k, neta, code, a, s_ag, s_beta, mu, c.tau, best);
        else
            chain_gemv<16>(d, kind, k, neta, code, a, s_ag, s_beta, mu, c.tau, best);
        best = dzg_block_best2(best);
        if (threadIdx.x == 0) {
            st_sc1(d.rx_r + blockIdx.x, best.r);
            st_sc1(d.rx_k + blockIdx.x, best.k);
            st_sc1(d.rx_h + blockIdx.x, best.h);
        }
        ts.mark(slot); // 3: FTRAN rows + candidates
        if (!chain_barrier(ctl, bar, gen)) return;
        ts.mark(slot); // 4: barrier
        DzgCand2 cw = dzg_cand2_none();
        for (int i = threadIdx.x; i < nwg; i += blockDim.x) {
            DzgCand2 o;
            o.r = ld_sc1(d.rx_r + i);
            o.k = ld_sc1(d.rx_k + i);
            o.h = ld_sc1(d.rx_h + i);
            cw = dzg_better2(cw, o);
        }
        cw = dzg_block_best2(cw);
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
    if (lead) {
        ctl->xp = d.x[p];
        ctl->xbp = d.xbar[p];
        ctl->leave_code = d.var_col[d.basis[p]];
        if (gen != c.bar_gen) ctl->bar_gen = gen;
    }
    // ---- BTRAN: v = row p of Binv on this workgroup's rows
    const int lcode = d.var_col[d.basis[p]];
    __syncthreads();
    if (threadIdx.x < R_) s_up[threadIdx.x] = (int)threadIdx.x < neta ? d.U[(long long)threadIdx.x * d.ldw + p] : 0.0;
    __syncthreads();
    int r0, r1;
    chain_rows(m, r0, r1);
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int slot = d.dslot[r];
        const double base = slot >= 0 ? d.binv[(long long)p * d.ldb + slot] : (lcode == -1 - r ? 1.0 : 0.0);
        double acc = 0.0;
        for (int t = 0; t < neta; ++t) acc = fma(s_up[t], d.W[(long long)t * d.ldw + r], acc);
        d.v[r] = base - acc;
    }
    ts.mark(slot); // primal 6 / dual 1: BTRAN row
    ts.done(slot);
}

// ---------------------------------------------------------------------------------
// k_chain_post: a dual step's ratio test and FTRAN (src/simplex.rs:324-325, :226-229), the pivot's
// books (fast_rows.h), pivot() x4 (:262-265, :410-421) with the eta append, and the first-pivot
// candidates of the next iteration (:423-437).  only_partials != 0: the candidates only.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(CH_THREADS) void k_chain_post(const DzgDev d, unsigned long long *bar,
                                                           const DzgPivotArgs pa, int only_partials,
                                                           int nrz, unsigned long long *dbg)
{
    ChainStamps ts;
    ts.start(only_partials ? nullptr : dbg);
    int slot = 2;
    __shared__ double s_ag[CH_AGCAP];
    __shared__ double s_beta[R_];
    __shared__ double s_dxp;
    DzgCtl *ctl = d.ctl;
    DzgCtl c = *ctl;
    if (c.status != DZG_RUNNING) return;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int m = d.m, q = d.q;
    unsigned long long gen = c.bar_gen;
    int r0, r1;
    chain_rows(m, r0, r1);
    // what the update needs (all block-uniform)
    int p = 0, r = 0, teta = 0, wzero = -1, del_ce = -1, del_last = -1, app_col = -1;
    double t = 0.0, s = 0.0, tbar = 0.0, sbar = 0.0, rdxp = 0.0, tau = c.tau;
    if (!only_partials) {
        const int neta = c.neta_cur, k = c.k_cur, kind = c.kind;
        const int ci = c.leave_code;
        const double xp = c.xp, xbp = c.xbp;
        p = c.leave_pos;
        int cj, edslot;
        double zr, zbr, dzr, dxp;
        bool books_wg;
        if (kind == DZG_STEP_DUAL) {
            const DzgCand2 cw = reduce_partials(d.rz_r, d.rz_k, d.rz_h, nrz);
            if (!fast_ratio_outcome(ctl, c, lead, cw, DZG_INFEASIBLE)) return; // :325
            slot = 3;
            ts.mark(slot); // 0: snapshot + ratio test
            r = cw.k;
            cj = d.var_col[d.nonbasis[r]];
            // read now: nobody rewrites them before the barrier below
            zr = d.z[r];
            zbr = d.zbar[r];
            dzr = d.dz[r];
            edslot = cj < 0 ? d.dslot[-1 - cj] : -1;
            if (lead) {
                ctl->enter_pos = r;
                ctl->enter_code = cj;
            }
            c.enter_pos = r;
            c.enter_code = cj;
            const double *a = cj < 0 ? nullptr : d.A + (long long)(cj - d.col0) * d.lda;
            chain_beta(d, neta, cj, a);
            chain_stage_ag(s_ag, k, cj, a, d.drow);
            ts.mark(slot); // 1: loads, beta, gather
            if (!chain_barrier(ctl, bar, gen)) return;
            ts.mark(slot); // 2: barrier
            if (threadIdx.x < R_) s_beta[threadIdx.x] = (int)threadIdx.x < neta ? ld_sc1(d.beta + threadIdx.x) : 0.0;
            __syncthreads();
            // dx_p by every workgroup itself (the arithmetic of the row's owner, so the same bits):
            // the step lengths then need nobody else's result
            const int k2 = (k + 1) & ~1;
            if (threadIdx.x < 64) {
                double acc;
                if (k > 512)
                    acc = fast_gemv_row<64>(p, m, k2, neta, d.binv, d.ldb, s_ag, d.U, d.ldw, s_beta, threadIdx.x);
                else
                    acc = fast_gemv_row<16>(threadIdx.x < 16 ? p : m, m, k2, neta, d.binv, d.ldb, s_ag, d.U,
                                            d.ldw, s_beta, threadIdx.x % 16);
                if (threadIdx.x == 0) s_dxp = fast_gemv_unit(acc, ci, cj, a);
            }
            DzgCand2 unused = dzg_cand2_none();
            if (k > 512)
                chain_gemv<64>(d, kind, k, neta, cj, a, s_ag, s_beta, 0.0, 0.0, unused);
            else
                chain_gemv<16>(d, kind, k, neta, cj, a, s_ag, s_beta, 0.0, 0.0, unused);
            __syncthreads();
            dxp = s_dxp;
            books_wg = p >= r0 && p < r1;
            ts.mark(slot); // 3: FTRAN rows + dx_p
        } else {
            r = c.enter_pos;
            cj = c.enter_code;
            zr = c.zr;
            zbr = c.zbar_r;
            dzr = d.dz[r];
            edslot = c.enter_dslot;
            dxp = d.dx[p];
            books_wg = blockIdx.x == 0;
            ts.mark(slot); // 0: snapshot + loads
        }
        const DzgPivotScalars ps = fast_pivot_scalars(xp, xbp, dxp, zr, zbr, dzr, neta, c.max_pivot_err);
        const bool appended = ci < 0, deleted = cj < 0;
        if (books_wg) {
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
        // a column about to be deleted is still being read by the FTRAN rows of slower workgroups
        if (kind == DZG_STEP_DUAL && deleted && !chain_barrier(ctl, bar, gen)) return;
        if (lead && gen != c.bar_gen) ctl->bar_gen = gen;
        t = ps.t;
        s = ps.s;
        tbar = ps.tbar;
        sbar = ps.sbar;
        teta = neta; // index of the eta appended now
        wzero = cj < 0 ? -1 - cj : -1;
        rdxp = 1.0 / dxp;
        if (c.tie_tol >= 0.0) { // the tolerance the books have just set
            const double adaptive = 64.0 * ps.max_err;
            tau = adaptive > c.tie_tol ? adaptive : c.tie_tol;
        }
        ts.mark(slot); // primal 1 / dual 4: step lengths (+ books in workgroup 0 / the owner)
    }
    // ---- update of this workgroup's rows and columns; candidates on the updated values
    const double inf = __builtin_inf();
    double *ut = d.U + (long long)teta * d.ldw;
    double *wt = d.W + (long long)teta * d.ldw;
    DzgCand2 bx = dzg_cand2_none(), bz = dzg_cand2_none();
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        double xi = d.x[i], xb = d.xbar[i];
        if (!only_partials) {
            const double dd = d.dx[i];
            const double a = t * dd, b = tbar * dd;
            xi = (i == p) ? t : xi - a;
            xb = (i == p) ? tbar : xb - b;
            d.x[i] = xi;
            d.xbar[i] = xb;
            ut[i] = (i == p ? dd - 1.0 : dd) * rdxp;
            wt[i] = (i == wzero) ? 0.0 : d.v[i];
            // Binv0's columns: a leaving slack appends e_p, an entering slack deletes its column
            // (the last one takes its place)
            double *row = d.binv + (long long)i * d.ldb;
            if (del_last >= 0) {
                const double last_val = (app_col == del_last) ? (i == p ? 1.0 : 0.0) : row[del_last];
                if (del_ce != del_last) row[del_ce] = last_val;
                row[del_last] = 0.0;
            } else if (app_col >= 0 && i == p) {
                row[app_col] = 1.0;
            }
        }
        if (xb > 0.0) {
            DzgCand2 cn;
            cn.r = dzg_div(-xi, xb);
            cn.k = i;
            cn.h = -inf;
            if (cn.r == cn.r) bx = dzg_better2(bx, cn);
        }
        if (fabs(xb) <= tau && !(xi > tau)) bx.h = inf;
    }
    const int qper = (q + (int)gridDim.x - 1) / (int)gridDim.x;
    const int q0 = (int)blockIdx.x * qper, q1 = q0 + qper < q ? q0 + qper : q;
    for (int kk = q0 + threadIdx.x; kk < q1; kk += blockDim.x) {
        double zk = d.z[kk], zb = d.zbar[kk];
        if (!only_partials) {
            const double dd = d.dz[kk];
            const double a = s * dd, b = sbar * dd;
            zk = (kk == r) ? s : zk - a;
            zb = (kk == r) ? sbar : zb - b;
            d.z[kk] = zk;
            d.zbar[kk] = zb;
        }
        if (zb > 0.0) {
            DzgCand2 cn;
            cn.r = dzg_div(-zk, zb);
            cn.k = kk;
            cn.h = -inf;
            if (cn.r == cn.r) bz = dzg_better2(bz, cn);
        }
        if (fabs(zb) <= tau && !(zk > tau)) bz.h = inf;
    }
    bx = dzg_block_best2(bx);
    bz = dzg_block_best2(bz);
    if (threadIdx.x == 0) {
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

void dzg_launch_chain_pre(const DzgDev &d, int grid, unsigned long long *bar,
                          unsigned long long *dbg, hipStream_t st)
{
    hipLaunchKernelGGL(k_chain_pre, dim3(grid), dim3(CH_THREADS), 0, st, d, bar, dbg);
}

void dzg_launch_chain_post(const DzgDev &d, int grid, unsigned long long *bar,
                           unsigned long long *dbg, int only_partials, int nrz, hipStream_t st)
{
    hipLaunchKernelGGL(k_chain_post, dim3(grid), dim3(CH_THREADS), 0, st, d, bar, dzg_pivot_args(d),
                       only_partials, nrz, dbg);
}

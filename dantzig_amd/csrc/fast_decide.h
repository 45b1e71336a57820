// fast_decide.h -- the decisions of one FAST iteration, shared by the dense-inverse kernels
// (k_fast.hip) and the sparse-basis kernels (k_sparse.hip): status() (src/simplex.rs:274-306), the
// outcome of a ratio test (:313, :325), and the near-tie gate around both.
#pragma once
#include "common.h"

__device__ __forceinline__ DzgCand2 reduce_partials(const double *__restrict__ pr,
                                                    const int *__restrict__ pk,
                                                    const double *__restrict__ ph, int count)
{
    DzgCand2 best = dzg_cand2_none();
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        DzgCand2 c;
        c.r = pr[i];
        c.k = pk[i];
        c.h = ph[i];
        best = dzg_better2(best, c);
    }
    return dzg_block_best2(best);
}

// ---------------------------------------------------------------------------------
// Near-tie gate.  `margin` is the smallest relative margin of the decisions a kernel has just
// taken (dzg_margin; absolute for the optimality test).  Inside the tolerance, stop mode ends
// the run BEFORE the pivot with DZG_NEAR_TIE (every workgroup takes the same decision from the
// same data, so all return together); count mode records it and carries on.  `first` marks the
// first decision site of an iteration, which opens the per-pivot record.
// `c` is the caller's snapshot of the control block; what the lead lane writes to the control
// block is applied to it too, so a kernel that takes several decisions in a row (k_chain.hip)
// carries the record on without reading the control block again.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool tie_gate(DzgCtl *ctl, DzgCtl &c, bool lead, double margin,
                                         bool first)
{
    const bool inside = !(margin > c.tau);
    if (inside && c.tie_mode == 1 && c.iter != c.tie_skip_iter) {
        if (lead) ctl->status = DZG_NEAR_TIE;
        return true;
    }
    const double new_margin = first ? margin : (margin < c.margin ? margin : c.margin);
    const int new_seen = (first ? 0 : c.tie_seen) | (inside ? 1 : 0);
    if (lead) {
        ctl->margin = new_margin;
        ctl->tie_seen = new_seen;
    }
    c.margin = new_margin;
    c.tie_seen = new_seen;
    return false;
}

// A terminal verdict (optimal / unbounded / infeasible) taken inside the tolerance executes no
// pivot, so fast_pivot_books never books it: count it here.
__device__ __forceinline__ void tie_book_terminal(DzgCtl *ctl, const DzgCtl &c, double margin)
{
    if (!(margin > c.tau)) {
        ctl->near_ties = c.near_ties + 1;
        if (c.first_near_tie < 0) ctl->first_near_tie = c.iter;
    }
    if (margin < c.min_margin) ctl->min_margin = margin;
}

// margin of a ratio test (find_second_pivot): argmax margin, and the winner itself must be
// clearly positive -- a ratio that is positive only by rounding is excluded by the reference's
// `ratio > 0.0` (src/simplex.rs:455); none found: trustworthy unless some ratio was not
__device__ __forceinline__ double ratio_margin(DzgCand2 c, double tau)
{
    const double inf = __builtin_inf();
    if (c.k < 0) return c.h == -inf ? inf : -1.0;
    if (!(c.r > tau)) return 0.0;
    return dzg_margin(c);
}

// status() on the two first-pivot winners cj (z side) and ci (x side).  Returns false when the
// calling kernel has nothing more to do (terminated, stopped at a near tie, budget spent);
// otherwise `kind` is the step and the control block holds kind, mu, enter_pos / leave_pos.
__device__ __forceinline__ bool fast_status(DzgCtl *ctl, DzgCtl &c, bool lead,
                                            const DzgCand2 &cj_in, const DzgCand2 &ci_in, double eps,
                                            int m, bool from_records, int &kind_out,
                                            double *mu_out = nullptr)
{
    const double inf = __builtin_inf();
    int kind = -1, verdict = DZG_RUNNING;
    double mu = 0.0;
    // a side whose best entry is a pseudo-candidate (dzg_first_pivot_entry: ybar zero to within
    // rounding, y not) has no candidate FAST trusts: it is empty, and which branch of status()
    // the reference takes there is not FAST's to say -- flagged
    DzgCand2 cj = cj_in, ci = ci_in;
    bool unsure = false;
    if (cj.k >= 0 && cj.r == -inf) {
        cj.k = -1;
        unsure = true;
    }
    if (ci.k >= 0 && ci.r == -inf) {
        ci.k = -1;
        unsure = true;
    }
    // margin of status(): the argmax of the side that is used, the primal-vs-dual comparison,
    // the optimality test (absolute: eps is an absolute threshold), and no untrustworthy
    // ratio on the side whose index is not used (its VALUE still enters the comparisons)
    double margin = inf;
    if (cj.k >= 0 && ci.k >= 0) {
        const double primal = ci.r, dual = cj.r;
        const double top = primal > dual ? primal : dual;
        // eps is an absolute threshold at rounding level itself: the test is inside the
        // tolerance when `top` is within eps/2 of it, or within what the health monitor says
        // FAST's rounding amounts to (a degenerate optimum has top = 0 up to that rounding
        // in FAST and exactly in the reference: both sides of the test agree)
        const double noise = 64.0 * c.max_pivot_err > c.drift_tau ? 64.0 * c.max_pivot_err : c.drift_tau;
        const double tau_opt = noise > 0.5 * eps ? noise : 0.5 * eps;
        margin = fabs(top - eps) > tau_opt ? inf : 0.0;
        if (c.tie_tol < 0.0) margin = inf;
        if (primal <= eps && dual <= eps) {
            verdict = DZG_OPTIMAL;
        } else {
            const double a = fabs(primal), b = fabs(dual), den = a > b ? a : b;
            const double cmp = den > 0.0 && den < inf ? fabs(primal - dual) / den
                                                      : (primal == dual ? 0.0 : inf);
            if (cmp < margin) margin = cmp;
            if (primal < dual) {
                kind = DZG_STEP_PRIMAL;
                mu = dual;
            } else {
                kind = DZG_STEP_DUAL;
                mu = primal;
            }
        }
        const double mj = kind == DZG_STEP_PRIMAL ? dzg_margin(cj) : (cj.h == inf ? -1.0 : inf);
        const double mi = kind == DZG_STEP_DUAL ? dzg_margin(ci) : (ci.h == inf ? -1.0 : inf);
        if (mj < margin) margin = mj;
        if (mi < margin) margin = mi;
    } else if (cj.k >= 0) {
        kind = DZG_STEP_PRIMAL;
        mu = cj.r;
        margin = dzg_margin(cj);
        if (ci.h == inf) margin = -1.0;
    } else if (ci.k >= 0) {
        kind = DZG_STEP_DUAL;
        mu = ci.r;
        margin = dzg_margin(ci);
        if (cj.h == inf) margin = -1.0;
    } else {
        verdict = DZG_PANIC;
        if (ci.h == inf || cj.h == inf) margin = -1.0;
    }
    if (unsure && c.tie_tol >= 0.0) margin = -1.0;
    if (verdict == DZG_RUNNING && c.iter >= c.iter_stop) {
        if (lead) ctl->status = DZG_ITER_LIMIT;
        return false;
    }
    if (verdict == DZG_RUNNING && m == 0) verdict = DZG_PANIC;
    if (tie_gate(ctl, c, lead, margin, true)) return false;
    if (verdict != DZG_RUNNING) {
        if (lead) {
            ctl->status = verdict;
            tie_book_terminal(ctl, c, margin);
        }
        return false;
    }
    if (lead) {
        ctl->kind = kind;
        ctl->mu = mu;
        ctl->enter_pos = kind == DZG_STEP_PRIMAL ? cj.k : -1;
        ctl->leave_pos = kind == DZG_STEP_DUAL ? ci.k : -1;
        if (from_records) ctl->use_record = 0;
    }
    c.kind = kind;
    c.mu = mu;
    c.enter_pos = kind == DZG_STEP_PRIMAL ? cj.k : -1;
    c.leave_pos = kind == DZG_STEP_DUAL ? ci.k : -1;
    kind_out = kind;
    if (mu_out) *mu_out = mu;
    return true;
}

// Outcome of a ratio test whose block partials reduced to `cw`: near-tie gate, then `none_status`
// (DZG_UNBOUNDED in a primal step, DZG_INFEASIBLE in a dual step) when no candidate survived.
__device__ __forceinline__ bool fast_ratio_outcome(DzgCtl *ctl, DzgCtl &c, bool lead,
                                                   const DzgCand2 &cw, int none_status)
{
    const double margin = ratio_margin(cw, c.tau);
    if (tie_gate(ctl, c, lead, margin, false)) return false;
    if (cw.k < 0) {
        if (lead) {
            ctl->status = none_status;
            // (the iteration's smallest margin: a flagged status() before a clean ratio test counts)
            tie_book_terminal(ctl, c, c.margin);
        }
        return false;
    }
    return true;
}

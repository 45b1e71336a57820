"""The premise of the FAST pricing passes that skip rows (csrc/k_price_kernels.h: k_price_rows for
a dense matrix, k_price_csc_rl for CSC): v = row p of B^-1, the vector the reference prices with
(`solve_for_dz`, src/simplex.rs:231-236: v = LU(B^T).solve(e_p), then `neg_t_dot`,
src/linalg.rs:199-207), is zero outside R -- the rows whose slack is nonbasic -- and the leaving
slack's own row, where it is 1.  FAST's v has these entries EXACTLY (its BTRAN takes them from the
structure: base 0 or 1, eta rows that are zero there); the reference's LU solve returns them up to
rounding -- a few 1e-16 -- which is what this CPU test pins along the oracle's pivot logs with the
oracle's own LU: the rows the FAST passes leave out carry nothing but that noise in the reference's
sums.  (STRICT, which reproduces the reference bit for bit, keeps every row.)"""
import numpy as np
import pytest

from oracle import oracle as ora
from tests.lp_families import make_lp


def _support_violations(a, b, c, max_pivots=120):
    sf = ora.stdform_from_dense(a, b, c)
    res = ora.simplex_solve(sf, max_iter=max_pivots)
    m, n = sf.m, sf.n
    ns = n - m
    full = ora.csc_to_dense(m, n, sf.col_ptr, sf.row_idx, sf.val)
    basis = list(sf.basis)
    bad, checked, with_slack_leaving = [], 0, 0
    for it, (kind, entering, leaving, _mu) in enumerate(res.pivots):
        p = basis.index(leaving)
        bt = np.ascontiguousarray(full[:, basis].T)
        e_p = np.zeros(m)
        e_p[p] = 1.0
        v = ora.lu_solve(bt, e_p)  # Matrix::factorize + LU::solve of the reference, restated
        basic_slack_rows = {var - ns for var in basis if var >= ns}
        rl = leaving - ns if leaving >= ns else -1
        scale = max(1.0, float(np.abs(v).max()))
        for r in basic_slack_rows:
            want = 1.0 if r == rl else 0.0
            if not abs(v[r] - want) <= 1e-11 * scale:
                bad.append((it, r, float(v[r]), want))
        checked += len(basic_slack_rows)
        with_slack_leaving += rl >= 0
        basis[p] = entering
    return bad, checked, with_slack_leaving, len(res.pivots)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_reference_v_is_zero_up_to_rounding_outside_the_rows_the_fast_passes_read(kind):
    total_checked = total_slack = total_pivots = 0
    for seed in range(4100 + kind, 4100 + kind + 24, 3):
        a, b, c = make_lp(seed, kind, 1, 30)
        bad, checked, with_slack, pivots = _support_violations(a, b, c)
        assert not bad, (seed, bad[:5])
        total_checked += checked
        total_slack += with_slack
        total_pivots += pivots
    assert total_pivots > 30 and total_checked > 300 and total_slack > 3

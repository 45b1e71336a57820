"""Parity at BASELINE.json's full size (8192 x 16384) through size-independent properties: the
CPU oracle needs minutes per pivot here, so the checks are linearity and cross-kernel agreement
of the pricing pass, exact sequential sums on sampled columns, invariants of a FAST run, and
STRICT (reference arithmetic on the GPU) as the arbiter of the first pivots."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M, NS, SEED = 8192, 16384, 1003


@pytest.fixture(scope="module")
def core():
    from dantzig_amd import core as c

    return c


@pytest.fixture(scope="module")
def lp_data(core):
    return core.gen_dense_lp(seed=SEED, m=M, n_struct=NS)


def test_pricing_pass_properties(core, lp_data):
    a, _, _ = lp_data
    rng = np.random.default_rng(7)
    v1, v2 = rng.uniform(-1, 1, M), rng.uniform(-1, 1, M)
    cols = np.arange(NS)
    d1 = core.neg_t_dot(a, cols, v1, kernel=core.PRICE_SEQ)
    d2 = core.neg_t_dot(a, cols, v2, kernel=core.PRICE_SEQ)
    d12 = core.neg_t_dot(a, cols, v1 + v2, kernel=core.PRICE_SEQ)
    scale = np.sqrt(M)
    assert np.abs(d12 - (d1 + d2)).max() <= 1e-12 * scale * 10          # linearity in v
    dw = core.neg_t_dot(a, cols, v1, kernel=core.PRICE_WAVE)
    assert np.abs(dw - d1).max() <= 1e-12 * scale * 10                   # both kernels agree
    sample = rng.choice(NS, 48, replace=False)
    ref = -(np.asarray(a)[:, sample].T @ v1)
    assert np.abs(d1[sample] - ref).max() <= 1e-12 * scale * 10          # vs numpy (pairwise sums)
    for j in sample[:6]:                                                  # exact sequential sums
        acc = 0.0
        col = np.asarray(a)[:, j]
        for i in range(M):
            acc = acc + col[i] * -v1[i]
        assert d1[j] == acc
    # unit columns and permutations of the column list
    perm = rng.permutation(NS)[:4096]
    mix = np.concatenate([perm, -1 - rng.integers(0, M, 64)])
    dm = core.neg_t_dot(a, mix, v1, kernel=core.PRICE_SEQ)
    assert np.array_equal(dm[:4096], d1[perm])
    assert np.array_equal(dm[4096:], 0.0 + -v1[-1 - mix[4096:]])


def test_fast_run_invariants(core, lp_data):
    a, b, c = lp_data
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, poll_interval=50) as s:
        assert s.run(600) == "iter_limit"
        res = s.result()
    assert res.iterations == 600 and len(res.pivots) == 600
    both = np.concatenate([res.basis, res.nonbasis])
    assert np.array_equal(np.sort(both), np.arange(NS + M))               # still a partition
    assert res.max_pivot_error < 1e-10
    # primal consistency: A x_struct + slack = b for the basic solution the engine carries
    xs = np.zeros(NS)
    slack = np.zeros(M)
    for pos, var in enumerate(res.basis):
        if var < NS:
            xs[var] = res.x[pos]
        else:
            slack[var - NS] = res.x[pos]
    basic_cols = res.basis[res.basis < NS]
    resid = np.asarray(a)[:, basic_cols] @ xs[basic_cols] + slack - b
    assert np.abs(resid).max() <= 1e-9
    # every pivot moved one nonbasic variable in and one basic variable out
    assert all(e != l for _, e, l, _ in res.pivots)
    # a column that entered is structural or slack consistently with the final basis size
    k = int((res.basis < NS).sum())
    assert 0 < k <= 600


def test_first_pivots_match_reference_arithmetic(core, lp_data):
    a, b, c = lp_data
    lp = core.CoreLP.from_inequality_form(a, b, c)
    strict = core.solve(lp, numerics=core.STRICT, max_iter=3)
    fast = core.solve(lp, numerics=core.FAST, max_iter=3)
    assert [(k, e, l) for k, e, l, _ in fast.pivots] == [(k, e, l) for k, e, l, _ in strict.pivots]
    assert np.allclose([p[3] for p in fast.pivots], [p[3] for p in strict.pivots], rtol=1e-12)


@pytest.mark.parametrize("seed,m,ns", [(1006, 4096, 8192), (7, 4096, 2048), (8, 512, 16384),
                                       (9, 3000, 3000)])
def test_whole_solve_is_certified_optimal_by_lapack(core, seed, m, ns):
    """Whole solves (4096 x 8192: about 180 000 pivots; tall, wide and square shapes), then strong
    duality on the host: the final basis alone must be primal and dual feasible under LAPACK's
    arithmetic and the engine's objective must agree to the north star's 1e-9.  (8192 x 16384: tools/full_solve.py,
    profiles/r01_full_solve_8192x16384.txt -- 514 893 pivots, gap 7e-13.)"""
    from tests.optimality import certificate

    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    res = core.solve(lp, log=False, numerics=core.FAST, poll_interval=256)
    assert res.status == "optimal"
    assert res.max_pivot_error < 1e-9
    cert = certificate(np.asarray(a), b, c, res.basis)
    scale = max(1.0, abs(cert["primal_obj"]))
    assert cert["primal_infeas"] <= 1e-9 and cert["dual_infeas"] <= 1e-9
    assert abs(cert["primal_obj"] - cert["dual_obj"]) <= 1e-9 * scale
    assert abs(res.objective - cert["primal_obj"]) <= 1e-9 * scale

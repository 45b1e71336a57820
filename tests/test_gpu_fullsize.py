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


def _oracle_fixture(kind, seed, m, ns):
    import json
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", f"oracle_{kind}_pivots_{seed}_{m}x{ns}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def test_first_pivots_are_the_cpu_oracles(core, lp_data):
    """The headline LP against pivots the CPU ORACLE holds (not a HIP-vs-HIP chain, VERDICT r2 item
    1): tests/golden/oracle_first_pivots_1003_8192x16384.json is the literal C restatement of the
    reference (two dense LUs of B and of B^T from scratch per pivot, ~7 minutes of one core each),
    oracle_blocked_pivots_... its blocked twin's longer log (bit-equal factorisation,
    tests/test_oracle_kats.py; the literal pivots are its head).  STRICT must take the literal log
    with mu bit for bit (the reference's strict first-wins argmax, src/simplex.rs:423-461, in the
    reference's arithmetic); FAST must take every pivot of the long log -- kind, entering, leaving
    -- with mu to 1e-9 relative (north star) and no decision inside the tie tolerance."""
    lit = _oracle_fixture("first", SEED, M, NS)
    assert lit is not None and len(lit["kind"]) >= 8
    long_ = _oracle_fixture("blocked", SEED, M, NS) or lit
    a, b, c = lp_data
    lp = core.CoreLP.from_inequality_form(a, b, c)
    n_strict = min(len(lit["kind"]), 10)
    strict = core.solve(lp, numerics=core.STRICT, max_iter=n_strict)
    assert [(k, e, l) for k, e, l, _ in strict.pivots] == list(
        zip(lit["kind"][:n_strict], lit["entering"][:n_strict], lit["leaving"][:n_strict]))
    assert [p[3] for p in strict.pivots] == lit["mu"][:n_strict]          # bit for bit
    n_fast = len(long_["kind"])
    fast = core.solve(lp, numerics=core.FAST, max_iter=n_fast)
    assert [(k, e, l) for k, e, l, _ in fast.pivots] == list(
        zip(long_["kind"], long_["entering"], long_["leaving"]))
    assert np.allclose([p[3] for p in fast.pivots], long_["mu"], rtol=1e-9, atol=0)
    assert fast.near_ties == 0


def test_strict_takes_fasts_pivots_deep_inside_the_solve(core, lp_data):
    """Beyond the pivots the CPU oracle holds, the arbiter is STRICT (the reference's arithmetic on the
    GPU) -- and STRICT from pivot 0 cannot reach the inside of a 515 000-pivot solve (0.64 s per pivot
    here).  So FAST's STATE is handed over: after 30 000 pivots of the benchmark LP (k = 1 400) its
    basis, x, xbar, z, zbar go to a STRICT solver (core.resumed_from), which takes 8 pivots from there
    while FAST carries on by itself: the same pivots, mu to 1e-11.  tools/strict_windows.py does the
    same with windows of 150 pivots down to pivot 510 000 (profiles/r04_strict_windows_8192x16384.txt)."""
    a, b, c = lp_data
    lp = core.CoreLP.from_inequality_form(a, b, c)
    start, window = 30000, 8
    with core.Solver(lp, numerics=core.FAST, poll_interval=50, log_capacity=start + window + 64) as s:
        assert s.run(start) == "iter_limit"
        r0 = s.result(log=False)
        assert s.run(window) == "iter_limit"
        fast = s.result().pivots[start:start + window]
    assert r0.iterations == start and len(fast) == window and r0.dense_columns > 1000
    strict = core.solve(core.resumed_from(lp, r0), numerics=core.STRICT, max_iter=window)
    assert [(k, e, l) for k, e, l, _ in strict.pivots] == [(k, e, l) for k, e, l, _ in fast]
    assert np.allclose([p[3] for p in strict.pivots], [p[3] for p in fast], rtol=1e-11, atol=0)


@pytest.mark.parametrize("k", [5000, 8192])
def test_refactorisation_of_the_headline_basis_sizes(core, lp_data, k):
    """Config 3 is named after its on-device LU refactor: a basis of 8192 rows with k structural
    columns (k = 8192: the whole basis is dense; 5000: what the solve reaches) is factorised -- every
    sub-panel regime (8 rows per thread, then 4), panel pairs, rank-128 MFMA updates, the slack-row
    product -- and the fresh inverse must compute the pivot element of 48 pivots the same way twice
    (FTRAN vs BTRAN + pricing, see tests/test_gpu_parity.py::test_refactorisation_at_every_panel_shape)."""
    a, b, c = lp_data
    basis = np.concatenate([np.arange(k), NS + np.arange(k, M)]).astype(np.int64)
    nonbasis = np.concatenate([np.arange(k, NS), NS + np.arange(k)]).astype(np.int64)
    lp = core.CoreLP(a=np.asarray(a), c=np.concatenate([c, np.zeros(M)]), basis=basis, nonbasis=nonbasis,
                     x=np.ones(M), z=-np.ones(NS))
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=16) as s:
        status = s.run(48)
        r = s.result(log=False)
    assert status == "iter_limit" and r.iterations == 48 and r.refactors == 1
    assert r.max_pivot_error < 1e-8, r.max_pivot_error


def test_first_pivots_of_4096_rows_are_the_cpu_oracles(core):
    """The same at 4096 x 8192 (seed 1006, the LP whose whole solve is certified by LAPACK below):
    40 pivots of the literal restatement (~45 s of one core each)."""
    seed, m, ns = 1006, 4096, 8192
    lit = _oracle_fixture("first", seed, m, ns)
    assert lit is not None and len(lit["kind"]) >= 16
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    n = len(lit["kind"])
    want = list(zip(lit["kind"], lit["entering"], lit["leaving"]))
    strict = core.solve(lp, numerics=core.STRICT, max_iter=n)
    assert [(k, e, l) for k, e, l, _ in strict.pivots] == want
    assert [p[3] for p in strict.pivots] == lit["mu"]
    fast = core.solve(lp, numerics=core.FAST, max_iter=n)
    assert [(k, e, l) for k, e, l, _ in fast.pivots] == want
    assert np.allclose([p[3] for p in fast.pivots], lit["mu"], rtol=1e-9, atol=0)
    assert fast.near_ties == 0


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


# ------------------------------------------------------------------ BASELINE config 4
# sparse 50 000 x 100 000, 50 nonzeros per column (0.1 %), generator G2 seed 1004, kept CSC
M4, NS4, PER_COL4, SEED4 = 50_000, 100_000, 50, 1004


@pytest.fixture(scope="module")
def sparse_lp(core):
    return core.gen_sparse_lp(SEED4, M4, NS4, PER_COL4)


def test_config4_csc_pricing_is_the_reference_sum(core, sparse_lp):
    """k_price_csc at config 4's size: dz of sampled columns equals, bit for bit, the
    reference's loop -- stored entries in ascending-row order, product and sum rounded
    separately (src/linalg.rs:199-207) -- written out in Python; the whole pass agrees with
    scipy's CSC product to rounding; linear in v; unit columns exact."""
    import scipy.sparse as sp

    cp, ri, val, _, _ = sparse_lp
    assert cp[-1] == NS4 * PER_COL4 and np.all(np.diff(cp) == PER_COL4)
    rng = np.random.default_rng(11)
    v1, v2 = rng.uniform(-1, 1, M4), rng.uniform(-1, 1, M4)
    cols = np.arange(NS4)
    d1 = core.neg_t_dot_csc(M4, cp, ri, val, cols, v1)
    for j in rng.choice(NS4, 200, replace=False):
        acc = 0.0
        for e in range(cp[j], cp[j + 1]):
            assert e == cp[j] or ri[e] > ri[e - 1]          # rows ascend inside a column
            acc = acc + val[e] * -v1[ri[e]]
        assert d1[j] == acc
    a = sp.csc_matrix((val, ri, cp), shape=(M4, NS4))
    assert np.abs(d1 + a.T @ v1).max() <= 1e-13 * PER_COL4
    d2 = core.neg_t_dot_csc(M4, cp, ri, val, cols, v2)
    d12 = core.neg_t_dot_csc(M4, cp, ri, val, cols, v1 + v2)
    assert np.abs(d12 - (d1 + d2)).max() <= 1e-13 * PER_COL4
    mix = np.concatenate([rng.permutation(NS4)[:5000], -1 - rng.integers(0, M4, 64)])
    dm = core.neg_t_dot_csc(M4, cp, ri, val, mix, v1)
    assert np.array_equal(dm[:5000], d1[mix[:5000]])
    assert np.array_equal(dm[5000:], 0.0 + -v1[-1 - mix[5000:]])


def test_config4_first_pivots_are_the_cpu_oracles(core, sparse_lp):
    """the first pivots of the sparse 50 000 x 100 000 LP as the CPU
    ORACLE takes them -- the blocked twin of the C restatement on the reference's own CscMatrix over
    all columns, the basis densified like the reference does (src/linalg.rs:236-238): two dense LUs
    of 50 000 rows, 2 x 20 GB, 75 minutes of six cores per pivot in the build container
    (tests/golden/oracle_blocked_pivots_1004_50000x100000_csc50.json, make_oracle_first_pivots.py
    --blocked --sparse-per-col 50).  FAST on the sparse-basis path (matrix CSC on the device,
    live-entry pricing) takes them pivot for pivot with mu to 1e-9; STRICT -- the reference's
    arithmetic on the GPU, the same two 50 000-row LUs per pivot -- bit for bit."""
    import json
    import os

    path = os.path.join(os.path.dirname(__file__), "golden",
                        f"oracle_blocked_pivots_{SEED4}_{M4}x{NS4}_csc{PER_COL4}.json")
    assert os.path.exists(path), path
    with open(path) as f:
        fx = json.load(f)
    n = len(fx["kind"])
    assert n >= 1 and (fx["seed"], fx["m"], fx["n_struct"]) == (SEED4, M4, NS4)
    want = list(zip(fx["kind"], fx["entering"], fx["leaving"]))
    cp, ri, val, b, c = sparse_lp
    lp = core.CoreLP.from_csc(M4, cp, ri, val, b, c)
    fast = core.solve(lp, numerics=core.FAST, max_iter=n)
    assert [(k, e, l) for k, e, l, _ in fast.pivots] == want
    assert np.allclose([p[3] for p in fast.pivots], fx["mu"], rtol=1e-9, atol=0)
    assert fast.near_ties == 0
    strict = core.solve(lp, numerics=core.STRICT, max_iter=n)
    assert [(k, e, l) for k, e, l, _ in strict.pivots] == want
    assert [p[3] for p in strict.pivots] == fx["mu"]                      # bit for bit


def test_config4_fast_run_invariants(core, sparse_lp):
    """1 000 pivots of FAST numerics on config 4 (matrix CSC on the device): basis / nonbasis stay
    a partition, the basic solution the engine carries satisfies A x_B + slack = b under scipy's
    arithmetic, the two computations of every pivot element agree, and the pricing pass moved
    the bytes the roofline is priced on (12 B per stored entry of a nonbasic column)."""
    import scipy.sparse as sp

    cp, ri, val, b, c = sparse_lp
    lp = core.CoreLP.from_csc(M4, cp, ri, val, b, c)
    with core.Solver(lp, numerics=core.FAST, poll_interval=50) as s:
        assert s.run(1000) == "iter_limit"
        res = s.result()
    assert res.iterations == 1000 and len(res.pivots) == 1000
    both = np.concatenate([res.basis, res.nonbasis])
    assert np.array_equal(np.sort(both), np.arange(NS4 + M4))
    assert res.max_pivot_error < 1e-10
    assert 0 < res.dense_columns <= 1000
    xs, slack = np.zeros(NS4), np.zeros(M4)
    for pos, var in enumerate(res.basis):
        if var < NS4:
            xs[var] = res.x[pos]
        else:
            slack[var - NS4] = res.x[pos]
    a = sp.csc_matrix((val, ri, cp), shape=(M4, NS4))
    assert np.abs(a @ xs + slack - b).max() <= 1e-9
    # pricing walks the live entries only (rows with a nonbasic slack: at most 1000 of them so far,
    # ~2 * PER_COL4 entries each) beside its per-column and per-position constants
    per_launch = res.price_bytes / res.iterations
    assert 20 * (NS4 - 1000) + 8 * M4 + 32 * NS4 <= per_launch
    assert per_launch <= 20 * NS4 + 8 * M4 + 32 * NS4 + 16 * 1000 * 4 * PER_COL4
    # a budgeted continuation resumes the same trajectory: same log as an uninterrupted run
    with core.Solver(lp, numerics=core.FAST, poll_interval=64) as s2:
        assert s2.run(300) == "iter_limit" and s2.run(700) == "iter_limit"
        again = s2.result()
    assert again.pivots == res.pivots


# ------------------------------------------------------------------ BASELINE config 5
# dense 32768 x 65536, generator G1 seed 1005 -- on ONE GPU (16 GiB of matrix, 288 GB of HBM)
M5, NS5, SEED5 = 32768, 65536, 1005


@pytest.fixture(scope="module")
def config5(core):
    return core.gen_dense_lp(seed=SEED5, m=M5, n_struct=NS5)


def test_config5_first_pivots_are_the_strict_log(core, config5):
    """The first 80 pivots of config 5 equal the committed log of STRICT numerics -- the
    reference's arithmetic -- for this LP (tests/golden/pivots_1005_32768x65536.json; STRICT needs
    12.7 s per pivot here, profiles/r01_strict_vs_fast_32768x65536_80pivots.txt), no decision
    came within the tie tolerance, and the run keeps the engine's invariants."""
    import hashlib
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "pivots_1005_32768x65536.json")) as f:
        fx = json.load(f)
    assert (fx["m"], fx["n_struct"], fx["seed"]) == (M5, NS5, SEED5)
    a, b, c = config5
    lp = core.CoreLP.from_inequality_form(a, b, c)
    res = core.solve(lp, numerics=core.FAST, max_iter=fx["pivots"])
    log = [(k, e, l) for k, e, l, _ in res.pivots]
    assert log[:20] == list(zip(fx["kind"], fx["entering"], fx["leaving"]))[:20]
    assert hashlib.sha256(repr(log).encode()).hexdigest() == fx["strict_sha256"]
    assert np.allclose([p[3] for p in res.pivots], fx["mu_fast"], rtol=1e-12, atol=0)
    assert res.near_ties == 0 and res.min_margin > 1e-9
    assert res.max_pivot_error < 1e-12
    assert np.array_equal(np.sort(np.concatenate([res.basis, res.nonbasis])), np.arange(NS5 + M5))


def test_config5_first_pivots_are_the_cpu_oracles(core, config5):
    """pivots of config 5 that the CPU ORACLE holds -- the blocked twin of the C
    restatement (oracle/dzg_oracle_blocked.c: Matrix::factorize with the same operations on every
    element in the same order, bit-equal factors, tests/test_oracle_kats.py) ran the first pivots of
    32768 x 65536 seed 1005 in the build container, 25 minutes of six cores each: two dense LUs of
    32768 rows per pivot, the matrix handed over in the oracle's dense-input format
    (tests/golden/oracle_blocked_pivots_1005_32768x65536.json, make_oracle_first_pivots.py
    --blocked --dense-input).  STRICT -- the reference's arithmetic on the GPU -- takes them with mu
    bit for bit, FAST pivot for pivot with mu to 1e-9; the committed HIP-STRICT log of 80 pivots
    (test above) starts with them."""
    import json
    import os

    fx = _oracle_fixture("blocked", SEED5, M5, NS5)
    assert fx is not None and len(fx["kind"]) >= 2
    n = len(fx["kind"])
    want = list(zip(fx["kind"], fx["entering"], fx["leaving"]))
    a, b, c = config5
    lp = core.CoreLP.from_inequality_form(a, b, c)
    n_strict = min(n, 2)                                                  # (12.7 s per STRICT pivot here)
    strict = core.solve(lp, numerics=core.STRICT, max_iter=n_strict)
    assert [(k, e, l) for k, e, l, _ in strict.pivots] == want[:n_strict]
    assert [p[3] for p in strict.pivots] == fx["mu"][:n_strict]           # bit for bit
    fast = core.solve(lp, numerics=core.FAST, max_iter=n)
    assert [(k, e, l) for k, e, l, _ in fast.pivots] == want
    assert np.allclose([p[3] for p in fast.pivots], fx["mu"], rtol=1e-9, atol=0)
    assert fast.near_ties == 0
    with open(os.path.join(os.path.dirname(__file__), "golden", "pivots_1005_32768x65536.json")) as f:
        hip = json.load(f)
    assert list(zip(hip["kind"], hip["entering"], hip["leaving"]))[:n] == want


def test_config5_column_sharded_over_8_ranks_partitioned(core, config5):
    """BASELINE config 5 as it is stated: 32768 x 65536 column-block PARTITIONED over 8 ranks (a
    rank holds its 8192 columns only -- 2.1 GB -- the entering column travels in the exchange
    records, 256 KiB + 64 B each), all 8 ranks in lockstep on this one GPU (the exchange is a copy
    kernel; 17 GB of matrix + 8 replicated 8.6-GB inverses).  Every rank must take the first 80
    pivots of the committed log and hold the single-GPU run's numbers bit for bit: mu of every
    pivot, x, xbar, the objective."""
    import hashlib
    import json
    import os

    from dantzig_amd.sharded import make_lockstep, run_lockstep

    with open(os.path.join(os.path.dirname(__file__), "golden", "pivots_1005_32768x65536.json")) as f:
        fx = json.load(f)
    a, b, c = config5
    lp = core.CoreLP.from_inequality_form(a, b, c)
    single = core.solve(lp, numerics=core.FAST, max_iter=fx["pivots"])
    solvers = make_lockstep(lp, 8, replicate=False, max_iter=fx["pivots"], poll_interval=16)
    try:
        assert all(s.record_doubles == 8 + M5 for s in solvers)  # the column travels in the record
        assert [(s.col_begin, s.col_end) for s in solvers] == [(r * 8192, (r + 1) * 8192) for r in range(8)]
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == single.status == "iter_limit"
    log = [(k, e, l) for k, e, l, _ in single.pivots]
    assert hashlib.sha256(repr(log).encode()).hexdigest() == fx["strict_sha256"]
    for res in results:
        assert res.pivots == single.pivots           # kind, entering, leaving AND mu, exactly
        assert np.array_equal(res.x, single.x) and np.array_equal(res.xbar, single.xbar)
        assert np.array_equal(res.basis, single.basis)
        assert res.objective == single.objective
        assert res.near_ties == single.near_ties == 0 and res.min_margin == single.min_margin
        assert res.max_pivot_error == single.max_pivot_error


def test_config5_row_sharded_over_8_ranks_from_a_warm_start(core, config5):
    """config 5 DEEP in its solve -- warm-started from a basis of 16 384 structural
    columns (core.warm_started; the compact inverse is 32768 x 16384, 4.3 GB) -- column-block
    PARTITIONED over 8 ranks with the basis side sharded by rows (opts.shard_rows: a rank keeps 4096
    rows of the inverse, 0.54 GB).  The ranks factorise the starting basis together (their shares of
    A[R, S] and A[L, S] summed over the ranks) and then take, bit for bit, the single-GPU solver's
    pivots from the same state: mu of every pivot, x, xbar, basis, objective, monitor."""
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = config5
    lp = core.warm_started(core.CoreLP.from_inequality_form(a, b, c), 16384)
    single = core.solve(lp, numerics=core.FAST, max_iter=48, poll_interval=16)
    assert single.status == "iter_limit" and single.refactors == 1 and single.dense_columns > 16300
    assert single.max_pivot_error < 1e-8
    solvers = make_lockstep(lp, 8, replicate=False, shard_rows=True, max_iter=48, poll_interval=16)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == "iter_limit"
    for res in results:
        assert res.refactors == 1
        assert res.pivots == single.pivots           # kind, entering, leaving AND mu, exactly
        assert np.array_equal(res.x, single.x) and np.array_equal(res.xbar, single.xbar)
        assert np.array_equal(res.basis, single.basis) and res.objective == single.objective
        assert res.max_pivot_error == single.max_pivot_error and res.min_margin == single.min_margin


def test_config5_pricing_pass_properties(core, config5):
    """The streaming kernel at 32768 rows (256 tiles per column): against float64 numpy on
    sampled columns, linear in v, independent of which other columns are priced with it."""
    a = np.asarray(config5[0])[:, :13108]           # a 3.4 GB block: 13 108 columns, 16 per wave
    rng = np.random.default_rng(12)
    v1, v2 = rng.uniform(-1, 1, M5), rng.uniform(-1, 1, M5)
    cols = np.arange(a.shape[1])
    d1 = core.neg_t_dot(a, cols, v1, kernel=core.PRICE_TREE)
    d2 = core.neg_t_dot(a, cols, v2, kernel=core.PRICE_TREE)
    d12 = core.neg_t_dot(a, cols, v1 + v2, kernel=core.PRICE_TREE)
    scale = np.sqrt(M5)
    assert np.abs(d12 - (d1 + d2)).max() <= 1e-12 * scale * 10
    pick = rng.choice(len(cols), 40, replace=False)
    ref = -(np.asarray(a)[:, cols[pick]].T @ v1)
    assert np.abs(d1[pick] - ref).max() <= 1e-12 * scale * 10
    few = core.neg_t_dot(a, cols[pick], v1, kernel=core.PRICE_TREE)   # 1 column per wave
    assert np.array_equal(few, d1[pick])
    ds = core.neg_t_dot(a, cols[pick], v1, kernel=core.PRICE_SEQ)     # reference-order sums
    assert np.abs(ds - d1[pick]).max() <= 1e-12 * scale * 10

"""Pins the CPU oracle against every known-answer vector the reference's own tests
hold for the hot path: src/linalg.rs:306-446 exactly,
src/simplex.rs:484-796 to 1e-12, and the hand trace of SURVEY Appendix B."""
import numpy as np
import pytest

from oracle import oracle as ora


# ------------------------------------------------------------------ src/linalg.rs KATs
def test_lu_factorization(kats):
    k = kats["linalg"]["lu_factorization"]
    lu, p = ora.lu_factorize(np.array(k["a"]))
    assert p.tolist() == k["p"]
    assert lu.ravel().tolist() == k["lu"]  # exact: LINPACK-style unpermuted L


def test_lu_solve(kats):
    for k in kats["linalg"]["lu_solve"]:
        assert ora.lu_solve(np.array(k["a"]), np.array(k["b"])).tolist() == k["x"]


def test_matrix_roundtrip(kats):
    a = np.array(kats["linalg"]["matrix_roundtrip"]["a"])
    cp, ri, v = ora.csc_from_dense(a)
    assert ora.csc_to_dense(2, 2, cp, ri, v).tolist() == a.tolist()


def test_csc_from_dense(kats):
    k = kats["linalg"]["csc_from_dense"]
    cp, ri, v = ora.csc_from_dense(np.array(k["a"]))
    assert ri.tolist() == k["row_idx"]
    assert cp.tolist() == k["col_ptr"]
    assert v.tolist() == k["data"]


def test_csc_column_and_collect(kats):
    k = kats["linalg"]["csc_column"]
    cp, ri, v = ora.csc_from_dense(np.array(k["a"]))
    for j, col in enumerate(k["columns"]):
        assert ora.csc_column(3, cp, ri, v, j).tolist() == col
    # collect_columns([1,2,0]): gathered column c equals source column cols[c];
    # checked through neg_t_dot with unit vectors (the only consumer on the hot path)
    cols = kats["linalg"]["csc_collect_columns"]["cols"]
    for r in range(3):
        e = np.zeros(3)
        e[r] = 1.0
        got = ora.neg_t_dot(cp, ri, v, cols, e)
        want = [-k["columns"][c][r] for c in cols]
        assert np.array_equal(got, np.array(want) + 0.0)


def test_dense_transpose(kats):
    k = kats["linalg"]["dense_transpose"]
    assert ora.matrix_t(np.array(k["a"])).tolist() == k["t"]


def test_neg_transpose_dot(kats):
    k = kats["linalg"]["neg_transpose_dot"]
    cp, ri, v = ora.csc_from_dense(np.array(k["a"]))
    assert ora.neg_t_dot(cp, ri, v, [0, 1, 2, 3], np.array(k["v"])).tolist() == k["out"]


# ------------------------------------------------------------------ src/simplex.rs KATs
def _solver_ids(kats_path="tests/golden/reference_kats.json"):
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, kats_path)) as f:
        return [k["name"] for k in json.load(f)["solver"]]


@pytest.mark.parametrize("name", _solver_ids())
def test_solver_kat(kats, name):
    k = next(s for s in kats["solver"] if s["name"] == name)
    res = ora.solve_model(k["model"], max_iter=10_000)
    exp = k["expect"]
    assert res.status == exp["status"]
    if exp["status"] == "optimal":
        assert abs(res.objective - exp["objective"]) <= 1e-12
        for got, want in zip(res.values, exp["values"]):
            assert abs(got - want) <= 1e-12


# ------------------------------------------------------------------ SURVEY Appendix B
def test_readme_lp_pivot_trace():
    """README LP lowered as the Python surface lowers it (min x+y-z st x+y+z == 1):
    core objective -x-y+z, rows x+y+z<=1 and -x-y-z<=-1.  Appendix B's hand trace."""
    nn = {"lb": 0.0, "ub": None}
    model = {
        "vars": [nn, nn, nn],
        "objective": {"terms": [[0, -1.0], [1, -1.0], [2, 1.0]], "constant": -0.0},
        "constraints": [
            {"terms": [[0, 1.0], [1, 1.0], [2, 1.0]], "b": 1.0},
            {"terms": [[0, -1.0], [1, -1.0], [2, -1.0]], "b": -1.0},
        ],
    }
    sf = ora.build_standard_form(model)
    assert (sf.m, sf.n) == (5, 11)
    assert sf.basis.tolist() == [6, 7, 8, 9, 10]
    assert sf.nonbasis.tolist() == [0, 1, 2, 3, 4, 5]
    assert sf.z.tolist() == [1.0, -1.0, 1.0, -1.0, -1.0, 1.0]
    assert sf.x.tolist() == [1.0, -1.0, 0.0, 0.0, 0.0]
    assert np.signbit(sf.x[2:]).all()  # -lb of a nonneg variable is -0.0
    res = ora.simplex_solve(sf)
    assert res.status == "optimal"
    assert [(p[0], p[1], p[2]) for p in res.pivots] == [
        (ora.DUAL, 4, 7), (ora.PRIMAL, 1, 8), (ora.PRIMAL, 3, 9), (ora.PRIMAL, 7, 6)]
    assert [p[3] for p in res.pivots] == [1.0, 1.0, 1.0, 1.0]
    assert res.basis.tolist() == [7, 4, 1, 3, 10]
    assert res.objective == 1.0
    vals = ora.solution_values(sf, res)
    assert vals.tolist() == [0.0, 0.0, 1.0]


# ------------------------------------------------------------------ IEEE corner cases
def test_ratio_test_corner_semantics():
    # +inf participates and wins; NaN, -inf, zero and negatives are dropped (App. A.9)
    y = np.array([1.0, -1.0, 1.0, -1.0, -1.0, 1.0])
    ybar = np.ones(6)
    dy = np.array([1.0, -1.0, 1.0, -1.0, 1.0, -1.0])
    assert ora.find_second_pivot(1.0, y, ybar, dy) == 4
    assert ora.find_second_pivot(1.0, np.array([-1.0]), np.ones(1), np.array([0.0])) == -1
    # first maximum wins, 0.0 and -0.0 tie (App. A.1)
    assert ora.find_first_pivot(np.array([0.0, -0.0, 0.0]), np.ones(3)) == 0
    assert ora.find_first_pivot(np.array([-1.0, -2.0, -2.0]), np.ones(3)) == 1
    assert ora.find_first_pivot(np.array([-5.0, -9.0]), np.array([0.0, -1.0])) == -1


def test_zero_pivot_column_is_skipped():
    # src/linalg.rs:117: an all-zero pivot column leaves the matrix untouched
    a = np.array([[0.0, 1.0], [0.0, 2.0]])
    lu, p = ora.lu_factorize(a)
    assert p.tolist() == [0]
    assert lu.tolist() == a.tolist()


def test_optimality_certificate_agrees_with_the_oracle():
    """tests/optimality.py (the LAPACK duality check the GPU suite relies on at sizes the oracle
    cannot reach) on LPs the oracle does solve: its optimal bases must pass, a non-optimal basis
    must not."""
    import numpy as np

    from dantzig_amd import core  # host-side generator only
    from oracle import oracle as ora
    from tests.optimality import certificate

    for seed, m, ns in [(3, 12, 30), (4, 40, 25), (5, 33, 33)]:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        a = np.asarray(a)
        res = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
        assert res.status == "optimal"
        cert = certificate(a, b, c, res.basis)
        scale = max(1.0, abs(res.objective))
        assert cert["primal_infeas"] <= 1e-9 and cert["dual_infeas"] <= 1e-9
        assert abs(cert["primal_obj"] - cert["dual_obj"]) <= 1e-9 * scale
        assert abs(cert["primal_obj"] - res.objective) <= 1e-9 * scale
        early = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=max(1, res.iterations // 2))
        bad = certificate(a, b, c, early.basis)
        assert bad["primal_infeas"] > 1e-9 or bad["dual_infeas"] > 1e-9


def test_integer_lp_fixture_is_the_oracle():
    """tests/golden/integer_lps_200_400.json (what the GPU suite holds AUTO numerics to) really is
    the oracle's outcome: a sample of its cases is recomputed here."""
    import json
    import os

    from tests.lp_families import log_digest, make_lp

    with open(os.path.join(os.path.dirname(__file__), "golden", "integer_lps_200_400.json")) as f:
        fx = json.load(f)
    assert len(fx["cases"]) >= 200 and {c["kind"] for c in fx["cases"]} == {1, 2}
    assert all(fx["min_m"] <= c["m"] < fx["max_m"] for c in fx["cases"])
    for case in fx["cases"][::29]:
        a, b, c = make_lp(case["seed"], case["kind"], fx["min_m"], fx["max_m"])
        r = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=fx["cap"])
        assert (r.status, r.iterations, log_digest(r.pivots)) == (case["status"], case["pivots"],
                                                                  case["sha256"])


# ------------------------------------------------------------------ the blocked twin of Matrix::factorize
def _bits_equal(x, y):
    x, y = np.asarray(x, float), np.asarray(y, float)
    return x.shape == y.shape and bool(np.all((x.view(np.uint64) == y.view(np.uint64))
                                              | (np.isnan(x) & np.isnan(y))))


@pytest.mark.parametrize("n", [1, 2, 3, 31, 33, 63, 64, 65, 66, 97, 128, 129, 130, 200, 257, 500, 777])
def test_blocked_lu_equals_the_literal_restatement_bit_for_bit(n, monkeypatch):
    """oracle/dzg_oracle_blocked.c applies Matrix::factorize (src/linalg.rs:88-128) block by block
    on several cores -- the same operations on every element in the same order.  Packed factors and
    pivot vector must equal ora_lu_factorize's bit for bit: continuous data, small integers (exact
    ties in the pivot search), all-zero columns inside and beyond the first panel (the silently
    skipped zero pivot, :117), 0/1 matrices (singular more often than not: NaN/inf factors)."""
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    rng = np.random.default_rng(1000 + n)
    for kind in range(4):
        if kind == 0:
            a = rng.uniform(-1, 1, (n, n))
        elif kind == 1:
            a = rng.integers(-2, 3, (n, n)).astype(np.float64)
        elif kind == 2:
            a = rng.uniform(-1, 1, (n, n))
            if n > 3:
                a[:, n // 3] = 0.0
                a[:, min(n - 1, 70)] = 0.0
        else:
            a = (rng.uniform(size=(n, n)) < 0.1).astype(np.float64) + np.eye(n) * (rng.uniform(size=n) < 0.7)
        lu0, p0 = ora.lu_factorize(a)
        lu1, p1 = ora.lu_factorize_blocked(a)
        assert p0.tolist() == p1.tolist(), (n, kind)
        assert _bits_equal(lu0, lu1), (n, kind)


def test_blocked_oracle_solve_equals_the_literal_one():
    """The whole iteration through the twin library (only the factorisation differs): same pivot
    log, mu, vectors and objective, bit for bit -- continuous and integer data."""
    from dantzig_amd import core
    from tests.lp_families import make_lp

    a, b, c = core.gen_dense_lp(seed=31, m=300, n_struct=640)
    cases = [(np.asarray(a), b, c, 150)] + [make_lp(s, 1 + s % 2, 40, 90) + (400,) for s in (5, 6, 7)]
    for a, b, c, cap in cases:
        sf = ora.stdform_from_dense(a, b, c)
        r0 = ora.simplex_solve(sf, max_iter=cap)
        r1 = ora.simplex_solve(sf, max_iter=cap, blocked=True)
        assert r0.status == r1.status and r0.iterations == r1.iterations
        assert r0.pivots == r1.pivots or _bits_equal([p[3] for p in r0.pivots], [p[3] for p in r1.pivots])
        assert [p[:3] for p in r0.pivots] == [p[:3] for p in r1.pivots]
        for name in ("x", "xbar", "z", "zbar"):
            assert _bits_equal(getattr(r0, name), getattr(r1, name)), name


def _first_pivot_fixture(kind, seed, m, ns):
    import json
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", f"oracle_{kind}_pivots_{seed}_{m}x{ns}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def test_pivot_fixtures_at_benchmark_size_agree():
    """BASELINE config 3 (8192 x 16384, seed 1003): the LITERAL restatement's first pivots
    (tests/golden/oracle_first_pivots_*.json, about 7 minutes of one core each) are the head of the
    blocked twin's longer log (oracle_blocked_pivots_*.json, 20 s each) -- kind, entering, leaving
    and every bit of mu.  Same for 4096 x 8192 where both exist."""
    checked = 0
    for seed, m, ns in [(1003, 8192, 16384), (1006, 4096, 8192)]:
        lit, blk = _first_pivot_fixture("first", seed, m, ns), _first_pivot_fixture("blocked", seed, m, ns)
        if lit is None or blk is None:
            continue
        n = min(len(lit["kind"]), len(blk["kind"]))
        assert n >= 2
        for key in ("kind", "entering", "leaving", "mu"):
            assert lit[key][:n] == blk[key][:n], (m, key)
        checked += 1
    assert checked >= 1


def test_dense_input_format_is_the_same_oracle():
    """ora_simplex with row_idx == NULL reads the structural block as a dense column-major array with
    implicit unit slack columns (oracle/dzg_oracle.h): the entries it visits are the CSC's -- exact
    zeros skipped, ascending rows -- so pivots, mu and the state vectors are bit-equal in the two
    formats (continuous data with planted zeros, small integers).  The format exists for fixtures at
    sizes whose 64-bit CSC does not fit the build container (config 5: 34 GB)."""
    import ctypes as C

    from dantzig_amd import core
    from tests.lp_families import make_lp

    cases = []
    for seed, m, ns in [(5, 40, 90), (6, 64, 64), (7, 33, 100)]:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        a = np.array(a)
        a[np.abs(a) < 0.05] = 0.0
        cases.append((a, b, c))
    cases += [make_lp(seed, 1, 10, 40) for seed in range(300, 306)]
    for a, b, c in cases:
        m, ns = a.shape
        want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=5000)
        val = np.ascontiguousarray(a.T)
        basis, nonbasis = np.arange(ns, ns + m, dtype=np.int64), np.arange(ns, dtype=np.int64)
        x, z = b.astype(np.float64).copy(), -c.astype(np.float64)
        xbar, zbar = np.ones(m), np.ones(ns)
        cc, cp = np.concatenate([c.astype(np.float64), np.zeros(m)]), np.zeros(1, dtype=np.int64)
        st = ora._Simplex(m, ns + m, ora._p(cp), None, ora._p(val), ora._p(cc), 0.0, ora._p(basis),
                          ora._p(nonbasis), ora._p(x), ora._p(xbar), ora._p(z), ora._p(zbar))
        log = (ora._Pivot * 5000)()
        it = C.c_int64(0)
        status = ora.lib().ora_simplex_solve(C.byref(st), C.c_int64(5000), C.byref(it), log, C.c_int64(5000))
        got = [(log[i].kind, log[i].entering, log[i].leaving, log[i].mu) for i in range(it.value)]
        assert (ora.STATUS[status], it.value) == (want.status, want.iterations)
        assert got == want.pivots
        assert np.array_equal(x, want.x) and np.array_equal(z, want.z) and np.array_equal(basis, want.basis)

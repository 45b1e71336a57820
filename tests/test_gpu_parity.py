"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact for indices (pivot log, basis) everywhere; bit-exact floats
in STRICT numerics and for the order-preserving pricing kernel; 1e-9 relative on the
objective in FAST numerics (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    from dantzig_amd import core as c

    return c


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(got, want, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    # zeros of either sign compare equal (do not assert on zero signs)
    same = (_bits(got) == _bits(want)) | ((got == 0.0) & (want == 0.0))
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {same.size} values differ"


# ------------------------------------------------------------------ lu_solve (src/linalg.rs)
@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 63, 64, 65, 66, 127, 128, 129, 130, 257, 513, 2100,
                               4200])
def test_lu_solve_bit_exact(core, n):
    rng = np.random.default_rng(100 + n)
    a = rng.uniform(-1, 1, (n, n))
    b = rng.uniform(-1, 1, n)
    x, lu, p = core.lu_solve(a, b)
    want_lu, want_p = ora.lu_factorize(a)
    assert p.tolist() == want_p.tolist()
    assert_bit_equal(lu, want_lu, "packed LU")
    assert_bit_equal(x, ora.lu_solve(a, b), "x")


def test_division_is_correctly_rounded(core):
    """IEEE-754 division as x86 (the reference) does it: the compiler's fp64 sequence for gfx950
    is one unit off on quotients within ~1e-32 of a rounding boundary; dzg_div repairs it.
    A 1 x 1 system is one division (src/linalg.rs:297)."""
    rng = np.random.default_rng(1)
    cases = [(1.3999999999999992, -0.9999999999999994), (2.849572429871236e-14, 1.5543122344752195e-14),
             (11.0, 6.0), (1.0, 3.0), (-7.0, 5.0), (1e300, 1e-5), (1e-300, 1e5), (0.0, 2.0)]
    cases += [(float(n) / 10.0, float(d) / 10.0) for n in range(1, 40) for d in (3, 7, 9, 10, 11, 13)]
    cases += [(float(x), float(y)) for x, y in zip(rng.uniform(-2, 2, 200), rng.uniform(0.5, 2, 200))]
    for num, den in cases:
        x, _, _ = core.lu_solve(np.array([[den]]), np.array([num]))
        want = np.float64(num) / np.float64(den)
        assert _bits(x)[0] == _bits(np.array([want]))[0], (num, den, float(x[0]), float(want))


def test_lu_kats_on_gpu(core, kats):
    k = kats["linalg"]["lu_factorization"]
    _, lu, p = core.lu_solve(np.array(k["a"]), np.ones(3))
    assert p.tolist() == k["p"] and lu.ravel().tolist() == k["lu"]
    for s in kats["linalg"]["lu_solve"]:
        x, _, _ = core.lu_solve(np.array(s["a"]), np.array(s["b"]))
        assert x.tolist() == s["x"]


def test_lu_integer_ties_and_zero_pivot(core):
    # integer data: exact ties in the pivot search, exact zeros, a zero pivot column
    rng = np.random.default_rng(5)
    for n in (4, 9, 33):
        a = rng.integers(-2, 3, (n, n)).astype(np.float64)
        a[:, 1] = 0.0 if n == 4 else a[:, 1]
        b = rng.integers(-3, 4, n).astype(np.float64)
        x, lu, p = core.lu_solve(a, b)
        want_lu, want_p = ora.lu_factorize(a)
        assert p.tolist() == want_p.tolist()
        assert_bit_equal(lu, want_lu)
        want_x = ora.lu_solve(a, b)
        ok = np.isfinite(want_x)
        assert_bit_equal(x[ok], want_x[ok])
        assert np.array_equal(np.isnan(x), np.isnan(want_x))


def test_lu_zero_pivot_beyond_the_first_panel(core):
    """An all-zero column in the second 64-column panel: the step is skipped (src/linalg.rs:117)
    inside the panel AND in the blocked updates of the columns to its right."""
    rng = np.random.default_rng(11)
    n = 200
    a = rng.uniform(-1, 1, (n, n))
    a[:, 70] = 0.0
    a[:, 130] = 0.0
    b = rng.uniform(-1, 1, n)
    x, lu, p = core.lu_solve(a, b)
    want_lu, want_p = ora.lu_factorize(a)
    assert p.tolist() == want_p.tolist()
    assert_bit_equal(lu, want_lu, "packed LU")
    want_x = ora.lu_solve(a, b)
    ok = np.isfinite(want_x)
    assert_bit_equal(x[ok], want_x[ok], "x")
    assert np.array_equal(np.isnan(x), np.isnan(want_x))


def test_lu_back_substitution_outside_lds():
    """Right-hand sides too long for LDS stay in the output vector: same bits.  The switch is
    read once per process, hence the child process."""
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from dantzig_amd import core\n"
        "from oracle import oracle as ora\n"
        "rng = np.random.default_rng(3)\n"
        "for n in (130, 300):\n"
        "    a = rng.uniform(-1, 1, (n, n)); b = rng.uniform(-1, 1, n)\n"
        "    x, lu, p = core.lu_solve(a, b)\n"
        "    assert np.array_equal(x.view(np.int64), ora.lu_solve(a, b).view(np.int64)), n\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                         env=dict(os.environ, DZG_LU_SMALL_LDS="1"), timeout=300)
    assert run.returncode == 0 and "ok" in run.stdout, run.stderr[-2000:]


# ------------------------------------------------------------------ neg_t_dot (src/linalg.rs)
@pytest.mark.parametrize("m,ns", [(1, 1), (3, 4), (7, 5), (127, 33), (128, 32), (129, 31),
                                  (300, 100), (1000, 257)])
def test_neg_t_dot_seq_bit_exact(core, m, ns):
    rng = np.random.default_rng(7 * m + ns)
    a = rng.uniform(-1, 1, (m, ns))
    a[rng.uniform(size=a.shape) < 0.2] = 0.0  # exact zeros are dropped by the reference's CSC
    v = rng.uniform(-1, 1, m)
    cols = np.concatenate([rng.permutation(ns), -1 - rng.integers(0, m, 5)])
    rng.shuffle(cols)
    full = np.concatenate([a, np.eye(m)], axis=1)
    cp, ri, val = ora.csc_from_dense(full)
    ocols = np.where(cols >= 0, cols, ns + (-1 - cols))
    want = ora.neg_t_dot(cp, ri, val, ocols, v)
    got = core.neg_t_dot(a, cols, v, kernel=core.PRICE_SEQ)
    assert_bit_equal(got, want, "dz (sequential-order kernel)")
    got_w = core.neg_t_dot(a, cols, v, kernel=core.PRICE_WAVE)
    assert np.allclose(got_w, want, rtol=1e-12, atol=1e-13 * m)


def test_neg_t_dot_kat(core, kats):
    k = kats["linalg"]["neg_transpose_dot"]
    for kern in (core.PRICE_SEQ, core.PRICE_WAVE):
        got = core.neg_t_dot(np.array(k["a"]), [0, 1, 2, 3], np.array(k["v"]), kernel=kern)
        assert got.tolist() == k["out"]


# ------------------------------------------------------------------ pivot rules (src/simplex.rs)
def test_first_and_second_pivot_match_oracle(core):
    rng = np.random.default_rng(11)
    for n in (1, 5, 64, 1000, 5000):
        for trial in range(4):
            y = rng.uniform(-1, 1, n)
            ybar = rng.uniform(-0.5, 1, n)
            dy = rng.uniform(-1, 1, n)
            if trial >= 2:  # heavy ties, signed zeros, zero denominators -> +-inf, nan
                y = rng.integers(-1, 2, n).astype(float)
                ybar = rng.integers(0, 2, n).astype(float)
                dy = rng.integers(-1, 2, n).astype(float)
                y[rng.uniform(size=n) < 0.2] = -0.0
            mu = float(rng.integers(0, 3)) if trial >= 2 else float(rng.uniform(0, 2))
            assert core.first_pivot(y, ybar) == ora.find_first_pivot(y, ybar)
            assert core.second_pivot(mu, y, ybar, dy) == ora.find_second_pivot(mu, y, ybar, dy)
    assert core.first_pivot(np.array([1.0]), np.array([0.0])) == -1
    assert core.second_pivot(1.0, np.array([-1.0]), np.ones(1), np.array([0.0])) == -1
    # the reference's reduce starts on the first surviving element: a NaN ratio there sticks
    # (nothing is > NaN), a NaN further down is skipped (src/simplex.rs:432-435)
    inf, nan = float("inf"), float("nan")
    for y, ybar in [([-inf, -3.0, -5.0], [inf, 1.0, 1.0]),      # inf/inf first: sticks at 0
                    ([-1.0, -inf, -5.0], [1.0, inf, 1.0]),       # NaN in the middle: skipped
                    ([2.0, nan, -1.0], [0.0, 1.0, 1.0]),         # first SURVIVING element is the NaN
                    ([nan] + [-float(k) for k in range(2000)], [1.0] * 2001),
                    ([-float(k) for k in range(2000)] + [nan], [1.0] * 2001)]:
        y, ybar = np.array(y), np.array(ybar)
        assert core.first_pivot(y, ybar) == ora.find_first_pivot(y, ybar)


# ------------------------------------------------------------------ whole solves
def _oracle_dense(core, seed, m, ns):
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    return core.CoreLP.from_inequality_form(a, b, c), want


def _log(res):
    return [(k, e, l) for k, e, l, _ in res.pivots]


@pytest.mark.parametrize("seed,m,ns", [(1, 4, 8), (2, 16, 24), (3, 32, 64), (4, 64, 128),
                                       (5, 96, 64), (6, 128, 256)])
def test_strict_solve_is_bit_identical(core, seed, m, ns):
    lp, want = _oracle_dense(core, seed, m, ns)
    got = core.solve(lp, numerics=core.STRICT)
    assert got.status == want.status == "optimal"
    assert got.iterations == want.iterations
    assert _log(got) == _log(want)
    assert [p[3] for p in got.pivots] == [p[3] for p in want.pivots]  # mu, bit for bit
    assert got.basis.tolist() == want.basis.tolist()
    assert got.nonbasis.tolist() == want.nonbasis.tolist()
    for name in ("x", "xbar", "z", "zbar"):
        assert_bit_equal(getattr(got, name), getattr(want, name), name)
    assert got.objective == want.objective


@pytest.mark.parametrize("seed,m,ns", [(11, 32, 64), (12, 64, 128), (13, 128, 256),
                                       (14, 200, 300), (15, 256, 512)])
def test_fast_solve_matches_pivot_sequence(core, seed, m, ns):
    lp, want = _oracle_dense(core, seed, m, ns)
    for kern in (core.PRICE_SEQ, core.PRICE_WAVE):
        got = core.solve(lp, numerics=core.FAST, price_kernel=kern)
        assert got.status == want.status == "optimal"
        assert _log(got) == _log(want), "pivot sequence differs from the reference's"
        assert got.basis.tolist() == want.basis.tolist()
        assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
        assert np.allclose(got.x, want.x, rtol=1e-9, atol=1e-9)


def test_budgeted_runs_resume(core):
    lp, want = _oracle_dense(core, 21, 64, 128)
    with core.Solver(lp, numerics=core.FAST, poll_interval=4) as s:
        steps = 0
        while s.run(7) == "iter_limit":
            steps += 1
            assert steps < 10_000
        got = s.result()
    assert got.status == "optimal" and _log(got) == _log(want)


def test_iteration_cap(core):
    lp, want = _oracle_dense(core, 22, 32, 64)
    got = core.solve(lp, numerics=core.STRICT, max_iter=5)
    assert got.status == "iter_limit" and got.iterations == 5
    assert _log(got) == _log(want)[:5]


# ------------------------------------------------------------------ reference solver KATs
def _names(kind):
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_kats.json")) as f:
        return [k["name"] for k in json.load(f)[kind]]


@pytest.mark.parametrize("name", _names("solver"))
def test_reference_solver_kats_through_level2(kats, name):
    """src/simplex.rs:484-796 through dzg_model_solve (builder + GPU core + solution)."""
    import ctypes as C

    from dantzig_amd import _ffi

    k = next(s for s in kats["solver"] if s["name"] == name)
    md = k["model"]
    V = len(md["vars"])
    f64, i64 = _ffi.f64, _ffi.i64
    has_lb = np.array([v["lb"] is not None for v in md["vars"]] + [0], dtype=np.int32)
    has_ub = np.array([v["ub"] is not None for v in md["vars"]] + [0], dtype=np.int32)
    lb = f64([v["lb"] or 0.0 for v in md["vars"]] + [0.0])
    ub = f64([v["ub"] or 0.0 for v in md["vars"]] + [0.0])
    ot = md["objective"]["terms"]
    ov, oc = i64([t[0] for t in ot] + [0]), f64([t[1] for t in ot] + [0.0])
    ptr_, cv, cc, cb = [0], [], [], []
    for con in md["constraints"]:
        cv += [t[0] for t in con["terms"]]
        cc += [t[1] for t in con["terms"]]
        ptr_.append(len(cv))
        cb.append(con["b"])
    cp, cva, cca, cba = i64(ptr_), i64(cv + [0]), f64(cc + [0.0]), f64(cb + [0.0])
    p = _ffi.ptr
    model = _ffi.Model(V, p(has_lb), p(has_ub), p(lb), p(ub), len(ot), p(ov), p(oc),
                       md["objective"]["constant"], len(cb), p(cp), p(cva), p(cca), p(cba))
    want = ora.solve_model(md)
    # AUTO resolves to STRICT at these sizes: small, integer-valued, badly scaled LPs with exact
    # ties are what STRICT numerics exists for (FAST is checked on the dense random family)
    for numerics in (_ffi.STRICT, _ffi.AUTO):
        values = np.zeros(V + 1)
        res = _ffi.ModelResult()
        res.values = p(values)
        opts = _ffi.default_opts(numerics=numerics)
        rc = _ffi.check(_ffi.lib().dzg_model_solve(C.byref(model), C.byref(opts), C.byref(res)),
                        "dzg_model_solve")
        assert _ffi.status_str(rc) == k["expect"]["status"] == want.status
        if want.status == "optimal":
            assert abs(res.objective - k["expect"]["objective"]) <= 1e-12
            assert np.allclose(values[:V], k["expect"]["values"], rtol=0, atol=1e-12)
            assert res.numerics_used == _ffi.STRICT
            assert res.objective == want.objective
            assert values[:V].tolist() == want.values.tolist()
            assert res.iterations == want.iterations


@pytest.mark.parametrize("name", _names("solver"))
def test_reference_solver_kats_through_the_full_csc_entry(core, kats, name):
    """dzg_core_solve_full_csc takes `Simplex`'s fields as the reference holds them
    (src/simplex.rs:84-112): ONE CscMatrix over all n columns, slack columns included, 64-bit
    indices, b / n / x / z -- here exactly what the oracle's Simplex::new leaves (STRICT: pivot
    log, mu, vectors and objective bit for bit; the final b / n / x / z come back in place)."""
    k = next(s for s in kats["solver"] if s["name"] == name)
    sf = ora.build_standard_form(k["model"])
    want = ora.simplex_solve(sf)
    if sf.m == 0:
        pytest.skip("no rows: the reference underflows n - 1 (covered by the edge-shape tests)")
    got = core.core_solve_full_csc(sf.m, sf.n, sf.col_ptr, sf.row_idx, sf.val, sf.c, sf.constant,
                                   sf.basis, sf.nonbasis, sf.x, sf.z, numerics=core.STRICT)
    assert got.status == want.status == k["expect"]["status"]
    assert _log(got) == _log(want)
    assert _same_bits([p[3] for p in got.pivots], [p[3] for p in want.pivots])
    assert got.basis.tolist() == want.basis.tolist() and got.nonbasis.tolist() == want.nonbasis.tolist()
    for name_ in ("x", "xbar", "z", "zbar"):
        assert _same_bits(getattr(got, name_), getattr(want, name_)), name_
    if want.status == "optimal":
        assert got.objective == want.objective


def test_full_csc_entry_on_dense_and_sparse_lps(core):
    """The same entry on G1 / G2 LPs written the reference's way ([A | I] in one CSC): the dense one
    goes to the device as a dense block, the sparse one stays CSC (sparse-basis path in FAST);
    a structural column that happens to be a unit vector is handled either way."""
    rng = np.random.default_rng(4)
    a, b, c = core.gen_dense_lp(seed=61, m=40, n_struct=70)
    a = np.array(a)
    a[:, 5] = 0.0
    a[7, 5] = 1.0                      # a structural unit column in a row that also has a slack
    cp, ri, val, bs, cs = core.gen_sparse_lp(62, 300, 700, 5)
    dense_sparse = np.zeros((300, 700))
    for j in range(700):
        dense_sparse[ri[cp[j]:cp[j + 1]], j] = val[cp[j]:cp[j + 1]]
    for a_, b_, c_, numerics in [(a, b, c, core.STRICT), (a, b, c, core.FAST),
                                 (dense_sparse, bs, cs, core.FAST)]:
        sf = ora.stdform_from_dense(a_, b_, c_)
        want = ora.simplex_solve(sf)
        got = core.core_solve_full_csc(sf.m, sf.n, sf.col_ptr, sf.row_idx, sf.val, sf.c, sf.constant,
                                       sf.basis, sf.nonbasis, sf.x, sf.z, numerics=numerics)
        assert got.status == want.status == "optimal"
        assert _log(got) == _log(want)
        assert got.basis.tolist() == want.basis.tolist()
        if numerics == core.STRICT:
            assert _same_bits(got.x, want.x) and got.objective == want.objective
        else:
            assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
    del rng


# ------------------------------------------------------------------ independent optimum (HiGHS)
def _highs_cases():
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "highs_objectives.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _highs_cases(), ids=lambda c: f"{c['m']}x{c['n_struct']}")
def test_fast_full_solve_objective_matches_highs(core, case):
    """Whole FAST solves at sizes the CPU oracle cannot finish, against the optimal value an
    independent solver found for the same seeded LP (tests/golden/make_highs_fixtures.py).
    Tolerance 1e-9 relative (BASELINE.json north_star)."""
    a, b, c = core.gen_dense_lp(seed=case["seed"], m=case["m"], n_struct=case["n_struct"])
    lp = core.CoreLP.from_inequality_form(a, b, c)
    got = core.solve(lp, numerics=core.FAST, log=False, poll_interval=64)
    assert got.status == "optimal"
    want = case["objective"]
    assert abs(got.objective - want) <= 1e-9 * max(1.0, abs(want)), (got.objective, want)
    assert got.max_pivot_error < 1e-6
    # the final point is primal feasible: A x_B <= b up to rounding
    xs = np.zeros(case["n_struct"])
    pos = got.basis < case["n_struct"]
    xs[got.basis[pos]] = got.x[pos]
    assert (np.array(a) @ xs - b).max() <= 1e-7 and xs.min() >= -1e-9


# ------------------------------------------------------------------ FAST refactorisation
@pytest.mark.parametrize("seed,m,ns,interval", [(51, 64, 128, 40), (52, 128, 256, 96),
                                                (53, 200, 300, 64), (54, 150, 450, 17)])
def test_fast_refactor_keeps_pivot_sequence(core, seed, m, ns, interval):
    """Rebuilding the basis inverse from scratch (blocked LU with partial pivoting, MFMA
    trailing updates, explicit inverse) every few pivots must not change a single pivot."""
    lp, want = _oracle_dense(core, seed, m, ns)
    got = core.solve(lp, numerics=core.FAST, refactor_interval=interval, poll_interval=8)
    assert got.status == want.status == "optimal"
    assert _log(got) == _log(want)
    assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
    assert got.max_pivot_error < 1e-9


@pytest.mark.parametrize("m,k", [(700, 1), (700, 63), (700, 64), (700, 65), (700, 127), (700, 128),
                                 (700, 129), (700, 200), (700, 449), (700, 700), (2100, 1100),
                                 (2100, 2100)])
def test_refactorisation_at_every_panel_shape(core, m, k):
    """The factorisation of a basis with exactly k structural columns -- one narrow panel, full
    panels, a pair with a narrow second panel, pairs followed by a single panel (k_refactor.hip) --
    checked on what it is for: 48 pivots on the fresh inverse with the pivot element computed
    twice, by FTRAN (rows of the inverse times the entering column) and by BTRAN + pricing (row p of
    the inverse against the matrix); they agree to rounding only if the inverse is the inverse.
    x and z are arbitrary (x > 0 > z: the method has work to do); the refactorisation does not
    look at them."""
    ns = 2 * m
    a, b, c = core.gen_dense_lp(seed=300 + k, m=m, n_struct=ns)
    basis = np.concatenate([np.arange(k), ns + np.arange(k, m)]).astype(np.int64)
    nonbasis = np.concatenate([np.arange(k, ns), ns + np.arange(k)]).astype(np.int64)
    lp = core.CoreLP(a=np.asarray(a), c=np.concatenate([c, np.zeros(m)]), basis=basis, nonbasis=nonbasis,
                     x=np.ones(m), z=-np.ones(ns))
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=16) as s:
        status = s.run(48)
        r = s.result(log=False)
        assert status == "iter_limit" and r.iterations == 48 and r.refactors == 1
        assert r.max_pivot_error < 1e-9, r.max_pivot_error
        s.refactor()  # and again on the basis 48 pivots later (k has moved by a few)
        status = s.run(48)
        r = s.result(log=False)
    assert status == "iter_limit" and r.iterations == 96 and r.refactors == 2
    assert r.max_pivot_error < 1e-9, r.max_pivot_error


def test_fast_refactor_on_demand_large(core):
    """Refactor in the middle of a larger solve (k in the hundreds, several LU panels) and
    finish: the optimum must still match the independent HiGHS value."""
    case = next(c for c in _highs_cases() if c["m"] == 1024)
    a, b, c = core.gen_dense_lp(seed=case["seed"], m=case["m"], n_struct=case["n_struct"])
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=64) as s:
        for _ in range(6):
            if s.run(2500) != "iter_limit":
                break
            s.refactor()
        while s.run(0) == "iter_limit":
            pass
        got = s.result(log=False)
    assert got.status == "optimal"
    assert abs(got.objective - case["objective"]) <= 1e-9 * max(1.0, abs(case["objective"]))
    assert got.max_pivot_error < 1e-9


# ------------------------------------------------------------------ sparse (CSC) input
def _oracle_sparse(core, seed, m, ns, per_col):
    cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
    # oracle: full CSC over all n columns (structural block + identity slacks)
    col_ptr = np.concatenate([cp, cp[-1] + 1 + np.arange(m)])
    row_idx = np.concatenate([ri.astype(np.int64), np.arange(m)])
    vals = np.concatenate([val, np.ones(m)])
    sf = ora.StdForm(m=m, n=ns + m, col_ptr=col_ptr, row_idx=row_idx, val=vals,
                     c=np.concatenate([c, np.zeros(m)]), constant=0.0,
                     basis=np.arange(ns, ns + m), nonbasis=np.arange(ns), x=b.copy(), z=-c)
    want = ora.simplex_solve(sf)
    return core.CoreLP.from_csc(m, cp, ri, val, b, c), want


@pytest.mark.parametrize("seed,m,ns,per_col", [(61, 24, 60, 3), (62, 60, 150, 4), (63, 100, 260, 5)])
def test_sparse_strict_bit_identical(core, seed, m, ns, per_col):
    lp, want = _oracle_sparse(core, seed, m, ns, per_col)
    got = core.solve(lp, numerics=core.STRICT)
    assert got.status == want.status
    assert _log(got) == _log(want)
    assert [p[3] for p in got.pivots] == [p[3] for p in want.pivots]
    for name in ("x", "xbar", "z", "zbar"):
        assert_bit_equal(getattr(got, name), getattr(want, name), name)
    assert got.objective == want.objective


@pytest.mark.parametrize("seed,m,ns,per_col", [(64, 60, 150, 4), (65, 150, 400, 5), (66, 256, 700, 6)])
def test_sparse_fast_matches_pivot_sequence(core, seed, m, ns, per_col):
    lp, want = _oracle_sparse(core, seed, m, ns, per_col)
    got = core.solve(lp, numerics=core.FAST, poll_interval=16)
    assert got.status == want.status
    assert _log(got) == _log(want)
    if want.status == "optimal":
        assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))


# ------------------------------------------------------------------ parity at benchmark-scale sizes
@pytest.mark.parametrize("seed,m,ns,pivots", [(1002, 1024, 2048, 60), (2002, 2048, 4096, 24)])
def test_fast_matches_strict_pivots_at_scale(core, seed, m, ns, pivots):
    """Sizes the CPU oracle cannot reach in test time: STRICT numerics (bit-identical to the
    oracle wherever both can run) is the arbiter, FAST must take the same first pivots."""
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    strict = core.solve(lp, numerics=core.STRICT, max_iter=pivots)
    fast = core.solve(lp, numerics=core.FAST, max_iter=pivots)
    assert strict.status == fast.status == "iter_limit"
    assert _log(fast) == _log(strict)
    assert np.allclose([p[3] for p in fast.pivots], [p[3] for p in strict.pivots], rtol=1e-9)
    assert np.allclose(fast.x, strict.x, rtol=1e-8, atol=1e-9)


# ------------------------------------------------------------------ outcomes and edge shapes
@pytest.mark.parametrize("numerics", ["strict", "fast"])
def test_unbounded_infeasible_and_degenerate_shapes(core, numerics):
    num = core.STRICT if numerics == "strict" else core.FAST
    # max x0 + x1  st  -x0 + x1 <= 1  (unbounded along x0), x >= 0
    lp = core.CoreLP.from_inequality_form(np.array([[-1.0, 1.0]]), [1.0], [1.0, 1.0])
    assert core.solve(lp, numerics=num).status == "unbounded"
    # x0 + x1 <= -1 with x >= 0 is empty
    lp = core.CoreLP.from_inequality_form(np.array([[1.0, 1.0]]), [-1.0], [1.0, -1.0])
    assert core.solve(lp, numerics=num).status == "infeasible"
    # already optimal at the slack basis: no pivot at all
    lp = core.CoreLP.from_inequality_form(np.array([[1.0, 2.0], [3.0, 1.0]]), [1.0, 2.0], [-1.0, -2.0])
    res = core.solve(lp, numerics=num)
    assert res.status == "optimal" and res.iterations == 0 and res.objective == 0.0
    # no structural column at all (only slacks): the reference takes the one-sided branch of
    # status() (src/simplex.rs:299-303, no optimality test), its dual ratio test over an empty
    # set finds nothing and it reports Infeasible -- a reference quirk the engine reproduces
    lp = core.CoreLP(a=np.zeros((3, 0)), c=np.zeros(3), basis=np.arange(3), nonbasis=np.zeros(0, np.int64),
                     x=np.array([1.0, 2.0, 3.0]), z=np.zeros(0))
    res = core.solve(lp, numerics=num)
    want = ora.simplex_solve(ora.stdform_from_dense(np.zeros((3, 0)), np.array([1.0, 2.0, 3.0]), np.zeros(0)))
    assert res.status == want.status == "infeasible" and res.iterations == 0
    # a single row and a single column entered at the core boundary (no x- column, no bound
    # row): after one pivot zbar <= 0 everywhere, status() takes its one-sided branch without
    # an optimality test (src/simplex.rs:299-303) and the reference answers Infeasible although
    # x = 2 is optimal -- the engine must answer what the reference answers
    lp = core.CoreLP.from_inequality_form(np.array([[2.0]]), [4.0], [3.0])
    res = core.solve(lp, numerics=num)
    want = ora.simplex_solve(ora.stdform_from_dense(np.array([[2.0]]), np.array([4.0]), np.array([3.0])))
    assert res.status == want.status == "infeasible" and res.objective == want.objective == 6.0
    assert _log(res) == _log(want)


def test_fast_warm_start_from_a_non_slack_basis(core):
    """A caller-supplied basis with structural columns: FAST numerics factorises it on the
    device first (blocked LU + MFMA) and then iterates; STRICT and the oracle start from the
    same state, so all three must produce the same pivot log."""
    a, b, c = core.gen_dense_lp(seed=81, m=60, n_struct=120)
    first = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=25)
    assert first.status == "iter_limit" and (first.basis < 120).sum() > 5
    m, ns = 60, 120
    full = np.concatenate([np.asarray(a), np.eye(m)], axis=1)
    xb = np.linalg.solve(full[:, first.basis], b)            # x = B^-1 b for the new start
    cc = np.concatenate([c, np.zeros(m)])
    y = np.linalg.solve(full[:, first.basis].T, cc[first.basis])
    zn = full[:, first.nonbasis].T @ y - cc[first.nonbasis]  # z = N^T B^-T c_B - c_N
    cp, ri, val = ora.csc_from_dense(full)
    sf = ora.StdForm(m=m, n=ns + m, col_ptr=cp, row_idx=ri, val=val, c=cc, constant=0.0,
                     basis=first.basis.copy(), nonbasis=first.nonbasis.copy(), x=xb, z=zn)
    want = ora.simplex_solve(sf)
    lp = core.CoreLP(a=np.asarray(a), c=cc, basis=first.basis, nonbasis=first.nonbasis, x=xb, z=zn)
    strict = core.solve(lp, numerics=core.STRICT)
    fast = core.solve(lp, numerics=core.FAST, poll_interval=8)
    assert strict.status == fast.status == want.status
    assert _log(strict) == _log(want)
    assert _log(fast) == _log(want)
    assert abs(fast.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))


@pytest.mark.parametrize("seed,m,ns,per_col,interval", [(67, 80, 200, 4, 23), (68, 256, 700, 6, 64)])
def test_sparse_fast_refactor_keeps_pivot_sequence(core, seed, m, ns, per_col, interval):
    """CSC input: the refactorisation gathers its k x k block and the basic-slack rows straight
    from the sparse columns; periodic rebuilds must not change a pivot."""
    lp, want = _oracle_sparse(core, seed, m, ns, per_col)
    got = core.solve(lp, numerics=core.FAST, refactor_interval=interval, poll_interval=8)
    assert got.status == want.status
    assert _log(got) == _log(want)
    if want.status == "optimal":
        assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
    assert got.max_pivot_error < 1e-9


def test_sparse_fast_warm_start(core):
    """Warm start from a non-slack basis with the matrix kept in CSC on the device."""
    m, ns, per_col = 70, 180, 4
    cp, ri, val, b, c = core.gen_sparse_lp(69, m, ns, per_col)
    col_ptr = np.concatenate([cp, cp[-1] + 1 + np.arange(m)])
    row_idx = np.concatenate([ri.astype(np.int64), np.arange(m)])
    vals = np.concatenate([val, np.ones(m)])
    cc = np.concatenate([c, np.zeros(m)])
    sf0 = ora.StdForm(m=m, n=ns + m, col_ptr=col_ptr, row_idx=row_idx, val=vals, c=cc, constant=0.0,
                      basis=np.arange(ns, ns + m), nonbasis=np.arange(ns), x=b.copy(), z=-c)
    first = ora.simplex_solve(sf0, max_iter=20)
    assert first.status == "iter_limit" and (first.basis < ns).sum() > 3
    full = np.zeros((m, ns + m))
    for j in range(ns + m):
        full[row_idx[col_ptr[j]:col_ptr[j + 1]], j] = vals[col_ptr[j]:col_ptr[j + 1]]
    xb = np.linalg.solve(full[:, first.basis], b)
    y = np.linalg.solve(full[:, first.basis].T, cc[first.basis])
    zn = full[:, first.nonbasis].T @ y - cc[first.nonbasis]
    sf = ora.StdForm(m=m, n=ns + m, col_ptr=col_ptr, row_idx=row_idx, val=vals, c=cc, constant=0.0,
                     basis=first.basis.copy(), nonbasis=first.nonbasis.copy(), x=xb, z=zn)
    want = ora.simplex_solve(sf)
    lp = core.CoreLP(a=None, c=cc, basis=first.basis, nonbasis=first.nonbasis, x=xb, z=zn,
                     col_ptr=cp, row_idx=ri, val=val)
    fast = core.solve(lp, numerics=core.FAST, poll_interval=8)
    assert fast.status == want.status
    assert _log(fast) == _log(want)


@pytest.mark.parametrize("seed", [91, 92, 93])
def test_a_solve_resumes_from_the_six_state_arrays(core, seed):
    """dzg_lp.xbar / zbar (Simplex.x_bar, z_bar): what a result hands back -- basis, nonbasis, x,
    xbar, z, zbar -- is the whole state of the reference's loop (src/simplex.rs:226-236 factorises
    from scratch every iteration), so a STRICT solve stopped after N pivots and resumed in a NEW
    solver takes the oracle's remaining pivots and ends on the oracle's bits; FAST (dense and the
    sparse-basis path: the basis is factorised on the device at creation) takes the same pivots."""
    a, b, c = core.gen_dense_lp(seed=seed, m=40, n_struct=70)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    stop = want.iterations // 2
    assert stop >= 5
    lp = core.CoreLP.from_inequality_form(a, b, c)
    part = core.solve(lp, numerics=core.STRICT, max_iter=stop)
    assert part.status == "iter_limit" and part.iterations == stop
    assert not (np.all(part.xbar == 1.0) and np.all(part.zbar == 1.0))
    rest = core.solve(core.resumed_from(lp, part), numerics=core.STRICT)
    assert rest.status == want.status
    assert _log(part) + _log(rest) == _log(want)
    assert np.array_equal(rest.basis, want.basis)
    for name in ("x", "xbar", "z", "zbar"):
        assert_bit_equal(getattr(rest, name), getattr(want, name), name)
    assert rest.objective == want.objective
    fast = core.solve(core.resumed_from(lp, part), numerics=core.FAST, poll_interval=8)
    assert fast.status == want.status and fast.refactors >= 1
    assert _log(part) + _log(fast) == _log(want)
    assert abs(fast.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
    # the sparse-basis path, stopped and resumed in FAST numerics
    lp_s, want_s = _oracle_sparse(core, seed, 70, 180, 4)
    half = core.solve(lp_s, numerics=core.FAST, max_iter=want_s.iterations // 2, poll_interval=8)
    assert half.status == "iter_limit" and half.dense_columns > 0
    tail = core.solve(core.resumed_from(lp_s, half), numerics=core.FAST, poll_interval=8)
    assert tail.status == want_s.status
    assert _log(half) + _log(tail) == _log(want_s)
    if want_s.status == "optimal":
        assert abs(tail.objective - want_s.objective) <= 1e-9 * max(1.0, abs(want_s.objective))


# ------------------------------------------------------------------ complete solves vs the oracle
def _oracle_log_fixtures():
    import glob

    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_*.npz")))


@pytest.mark.parametrize("path", _oracle_log_fixtures(), ids=os.path.basename)
def test_whole_solve_follows_the_oracle_pivot_log(core, path):
    """Complete solves at sizes where the CPU oracle needs minutes to an hour (computed once,
    tests/golden/make_oracle_pivot_logs.py): FAST numerics must take the oracle's pivots -- kind,
    entering and leaving variable -- from the first to the last, end in its basis, and reach its
    objective to 1e-9."""
    fx = np.load(path)
    seed, m, ns = int(fx["seed"]), int(fx["m"]), int(fx["n_struct"])
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    got = core.solve(lp, numerics=core.FAST, poll_interval=64)
    assert got.status == str(fx["status"]) == "optimal"
    assert got.iterations == int(fx["iterations"])
    kinds = np.array([p[0] for p in got.pivots])
    enter = np.array([p[1] for p in got.pivots])
    leave = np.array([p[2] for p in got.pivots])
    assert np.array_equal(kinds, fx["kind"])
    assert np.array_equal(enter, fx["entering"]) and np.array_equal(leave, fx["leaving"])
    assert np.array_equal(got.basis, fx["basis"])
    mu = np.array([p[3] for p in got.pivots])
    # x and xbar are UPDATED vectors in the reference too (src/simplex.rs:262-265): after thousands
    # of pivots both sides carry their own rounding history in mu = -x/xbar (measured with
    # tools/mu_diag.py: 6e-9 after 7 692 pivots, 3e-8 after 21 642, unchanged by refactorising
    # every few hundred pivots), while every decision agrees
    assert np.all(np.abs(mu - fx["mu"]) <= 1e-7 * np.maximum(1.0, np.abs(fx["mu"])))
    assert abs(got.objective - float(fx["objective"])) <= 1e-9 * max(1.0, abs(float(fx["objective"])))


# ------------------------------------------------------------------ degenerate / integer LPs
def _fuzz_lp(core, case, kind=None):
    rng = np.random.default_rng(case)
    m, ns = int(rng.integers(1, 70)), int(rng.integers(1, 140))
    kind = case % 3 if kind is None else kind
    if kind == 0:
        a, b, c = core.gen_dense_lp(seed=case, m=m, n_struct=ns)
        return np.array(a), b, c
    if kind == 1:  # small integers, many zeros: exact ties in both pivot rules
        return (rng.integers(-3, 4, (m, ns)).astype(np.float64),
                rng.integers(-2, 9, m).astype(np.float64), rng.integers(-4, 5, ns).astype(np.float64))
    return ((rng.uniform(size=(m, ns)) < 0.3).astype(np.float64),  # 0/1 matrix: degenerate vertices
            rng.integers(0, 4, m).astype(np.float64), rng.integers(-1, 6, ns).astype(np.float64))


def _same_bits(x, y):
    x, y = np.asarray(x, float), np.asarray(y, float)
    return x.shape == y.shape and bool(np.all((_bits(x) == _bits(y)) | ((x == 0) & (y == 0))
                                              | (np.isnan(x) & np.isnan(y))))


def test_strict_follows_the_oracle_through_degenerate_lps(core):
    """Small-integer and 0/1 LPs: exact ties in the ratio tests, zero pivots, and the paths on which
    the reference's own arithmetic breaks down (0/0 -> NaN states, `safe_divide` asserts, false
    "unbounded" verdicts -- tools/fuzz_parity.py).  STRICT must go wherever the reference goes:
    same status, same pivots, same mu and the same vectors bit for bit, NaNs in the same places."""
    outcomes = set()
    # 2259 (as a 0/1 LP): the solve hits 1.3999999999999992 / -0.9999999999999994, a quotient so
    # close to a rounding boundary that the stock fp64 division sequence of gfx950 misses it
    # 4251 (as a 0/1 LP): a solve through a singular basis leaves x = -inf, xbar = +inf, the first
    # pivot rule starts on inf/inf = NaN and the reference ends in a dual step with mu = NaN
    for case, kind in [(c, None) for c in range(90)] + [(2259, 2), (4251, 2)]:
        a, b, c = _fuzz_lp(core, case, kind)
        want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=5000)
        got = core.solve(core.CoreLP.from_inequality_form(a, b, c), numerics=core.STRICT, max_iter=5000)
        outcomes.add(want.status)
        assert got.status == want.status, case
        assert _log(got) == _log(want), case
        assert _same_bits([p[3] for p in got.pivots], [p[3] for p in want.pivots]), case
        for name in ("x", "xbar", "z", "zbar"):
            assert _same_bits(getattr(got, name), getattr(want, name)), (case, name)
    assert {"optimal", "unbounded", "panic", "infeasible"} <= outcomes


def _csc_lp(core, a, b, c):
    """The same LP with its structural block in CSC (zeros dropped, src/linalg.rs:254-270)."""
    cp, ri, val = ora.csc_from_dense(np.asarray(a, dtype=np.float64))
    return core.CoreLP.from_csc(a.shape[0], cp, ri, val, b, c)


def test_neg_t_dot_skips_stored_zeros_as_the_reference_csc_does(core):
    """The reference's CSC holds no exact zeros (src/linalg.rs:254-270), so neg_t_dot
    (src/linalg.rs:199-207) never forms 0 * -v[i]; the device keeps dense columns, zeros included.
    With v = +/-inf or NaN in a row where a column holds a zero the reference SKIPS the entry and a
    plain product would give NaN: the sequential-order kernel must skip it too."""
    rng = np.random.default_rng(99)
    for m, ns in [(5, 7), (130, 40), (300, 33)]:
        a = rng.integers(-3, 4, (m, ns)).astype(np.float64)
        a[rng.uniform(size=a.shape) < 0.4] = 0.0
        a[0, 0] = -0.0  # a negative zero is dropped as well (`!= 0.0` is false for it)
        v = rng.integers(-4, 5, m).astype(np.float64)
        bad = rng.permutation(m)[:3]
        v[bad] = [np.inf, -np.inf, np.nan]
        a[np.ix_(bad, np.arange(ns))] *= rng.uniform(size=(3, ns)) < 0.3  # mostly zeros in those rows
        cols = np.concatenate([np.arange(ns), -1 - bad[:3]])
        full = np.concatenate([a, np.eye(m)], axis=1)
        cp, ri, val = ora.csc_from_dense(full)
        ocols = np.where(cols >= 0, cols, ns + (-1 - cols))
        want = ora.neg_t_dot(cp, ri, val, ocols, v)
        got = core.neg_t_dot(a, cols, v, kernel=core.PRICE_SEQ)
        assert _same_bits(got, want), (m, ns, np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want)))))
        # the case is not vacuous: columns that are finite although v is not (a plain product over
        # the dense column would make every one of them NaN), and columns that are not finite
        assert np.isfinite(want[:ns]).any() and (~np.isfinite(want[:ns])).any()
        with np.errstate(invalid="ignore"):
            assert not np.isfinite(a.T @ v).any()
        # the CSC kernel (stored entries only) is the same function
        gcsc = core.neg_t_dot_csc(m, cp[:ns + 1], ri[:cp[ns]], val[:cp[ns]], cols, v)
        assert _same_bits(gcsc, want), (m, ns)


@pytest.mark.parametrize("seed", [1932, 2656, 4366, 4538])
def test_strict_follows_the_oracle_where_zero_meets_a_non_finite_v(core, seed):
    """Small-integer LPs on which the reference reaches a non-finite v (BTRAN through a singular
    basis) while nonbasic columns hold zeros in those rows: the reference panics after 6 / 4 / 9 / 8
    pivots (`safe_divide`'s assert); with 0 * inf = NaN in the pricing pass the outcome would be
    `infeasible` instead (`make_lp(seed, 1, 1, 25)`)."""
    from tests.lp_families import make_lp

    a, b, c = make_lp(seed, 1, 1, 25)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=5000)
    assert want.status == "panic"
    for lp in (core.CoreLP.from_inequality_form(a, b, c), _csc_lp(core, a, b, c)):
        got = core.solve(lp, numerics=core.STRICT, max_iter=5000)
        assert got.status == want.status, seed
        assert _log(got) == _log(want), seed
        assert _same_bits([p[3] for p in got.pivots], [p[3] for p in want.pivots]), seed
        for name in ("x", "xbar", "z", "zbar"):
            assert _same_bits(getattr(got, name), getattr(want, name)), (seed, name)


def test_whole_solve_from_csc_follows_the_oracle_pivot_log(core):
    """The 512 x 1024 LP of the committed oracle log handed over as CSC (every entry stored): the
    sparse device path -- CSC pricing in the reference's order, entering columns densified into
    records -- takes the same 7 692 pivots."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_2001_512x1024.npz"))
    m, ns = int(fx["m"]), int(fx["n_struct"])
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=m, n_struct=ns)
    cp, ri, val = ora.csc_from_dense(np.asarray(a))
    lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
    got = core.solve(lp, numerics=core.FAST, poll_interval=64)
    assert got.status == "optimal" and got.iterations == int(fx["iterations"])
    assert np.array_equal(np.array([p[1] for p in got.pivots]), fx["entering"])
    assert np.array_equal(np.array([p[2] for p in got.pivots]), fx["leaving"])
    assert np.array_equal(got.basis, fx["basis"])


# ------------------------------------------------------------------ FAST near-tie arbitration
def _integer_fixture():
    import json

    with open(os.path.join(os.path.dirname(__file__), "golden", "integer_lps_200_400.json")) as f:
        return json.load(f)


def test_auto_follows_the_oracle_on_integer_lps(core):
    """100 small-integer and 0/1 LPs of 200-400 rows (every other one of the 200 the fixture holds:
    the suite's time budget; profiles/r02_fuzz_200_integer_lps_200_400_rows.txt has all of them) (exact ties, 0/0 and x/0 ratios: the data on
    which FAST numerics used to leave the reference's path without a signal) through
    dzg_core_solve with AUTO numerics: status, pivot count and pivot log (sha256 of the
    (kind, entering, leaving) triples) equal the committed CPU-oracle outcome on every one.
    AUTO = FAST that stops at the first decision within rounding of a tie, then STRICT from the
    first pivot (src/simplex.rs:423-461 is a strict first-wins argmax: only the reference's own
    arithmetic can arbitrate a tie)."""
    from tests.lp_families import log_digest, make_lp

    fx = _integer_fixture()
    bad, strict = [], 0
    for case in fx["cases"][::2]:  # (every other case of the fixture: the suite's time budget)
        a, b, c = make_lp(case["seed"], case["kind"], fx["min_m"], fx["max_m"])
        assert a.shape == (case["m"], case["ns"])
        lp = core.CoreLP.from_inequality_form(a, b, c)
        r = core.core_solve(lp, numerics=core.AUTO, max_iter=fx["cap"], log_cap=fx["cap"])
        strict += r.numerics == "strict"
        if (r.status, r.iterations, log_digest(r.pivots)) != (case["status"], case["pivots"],
                                                              case["sha256"]):
            bad.append((case["seed"], r.status, case["status"], r.iterations, case["pivots"],
                        r.numerics))
    assert not bad, bad
    assert strict > 75  # these LPs are full of ties: nearly all must have been handed to STRICT


def test_a_side_without_a_trustworthy_candidate_is_flagged(core):
    """Found by the fuzz (seed 40037 of tools/fuzz_parity.py, round 4): max 3 x0 + 4 x1 - 4 x2 over two
    integer rows.  After two pivots the reference holds xbar = (0, -0.5) with x = (8, 1): no xbar is
    positive, find_first_pivot finds nothing on the x side, and status() takes its one-sided branch --
    no optimality test (src/simplex.rs:299-303) -- into a primal step that ends Unbounded.  FAST's xbar_0
    is the same zero up to the sign of a rounding error; when that error is positive the entry used to
    be a candidate (ratio -8 / 1e-17), status() took the two-sided branch and answered Optimal, with no
    near tie flagged.  An entry whose xbar is zero to within rounding while x is not now stands aside
    (dzg_first_pivot_entry) and a side left without a real candidate is flagged: FAST follows the
    reference into the one-sided branch and says so."""
    a = np.array([[-3.0, -2.0, -3.0], [1.0, -2.0, 3.0]])
    b, c = np.array([6.0, -2.0]), np.array([-3.0, -4.0, 4.0])
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    assert want.status == "unbounded" and want.iterations == 2
    lp = core.CoreLP.from_inequality_form(a, b, c)
    strict = core.solve(lp, numerics=core.STRICT)
    assert strict.status == "unbounded" and _log(strict) == _log(want)
    for seven in (0, 1):
        fast = core.solve(lp, numerics=core.FAST, seven_launches=seven)
        assert fast.status == "unbounded" and _log(fast) == _log(want), (seven, fast.status)
        assert fast.near_ties >= 1 and fast.first_near_tie == 2  # the status() after the second pivot
    import scipy.sparse as sp

    acsc = sp.csc_matrix(a)
    lpc = core.CoreLP.from_csc(2, acsc.indptr, acsc.indices, acsc.data, b, c)
    fast = core.solve(lpc, numerics=core.FAST)
    assert fast.status == "unbounded" and _log(fast) == _log(want) and fast.near_ties >= 1


def test_auto_strict_resolve_runs_against_a_wall_clock_budget(core):
    """AUTO re-solves an LP that met a near tie in STRICT (3-57 ms per pivot): with a budget of one
    second an integer LP of 333 rows whose solve needs thousands of pivots (seed 9000 of the fixture
    is still running at its cap of 120) must not sit in STRICT -- it comes back from FAST with the
    near ties counted (numerics 'fast', near_ties > 0); with the default budget the same call is the
    STRICT answer."""
    from tests.lp_families import make_lp

    a, b, c = make_lp(9000, 1, 200, 400)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    quick = core.core_solve(lp, numerics=core.AUTO, max_iter=100000, log_cap=16, auto_strict_budget_s=1)
    assert quick.numerics == "fast" and quick.near_ties > 0 and quick.first_near_tie >= 0
    full = core.core_solve(lp, numerics=core.AUTO, max_iter=120, log_cap=120)
    assert full.numerics == "strict" and full.iterations == 120


def test_fast_leaves_the_oracle_path_only_where_it_flagged_a_near_tie(core):
    """FAST with near ties COUNTED (no stop) on 150 small LPs of all three families: wherever its
    pivot log or verdict differs from the oracle's, a near tie was flagged at or before the first
    differing pivot; on continuous data nothing is flagged and nothing differs."""
    from tests.lp_families import log3, make_lp

    unflagged, flagged_continuous = [], []
    for seed in range(7000, 7150):
        kind = seed % 3
        a, b, c = make_lp(seed, kind, 1, 60)
        want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=5000)
        lp = core.CoreLP.from_inequality_form(a, b, c)
        f = core.solve(lp, numerics=core.FAST, max_iter=5000, poll_interval=8)
        got, wlog = log3(f.pivots), log3(want.pivots)
        if kind == 0 and (f.near_ties or got != wlog or f.status != want.status):
            flagged_continuous.append((seed, f.near_ties, f.status, want.status))
        if got != wlog or f.status != want.status:
            d = next((i for i, (p, q) in enumerate(zip(got, wlog)) if p != q),
                     min(len(got), len(wlog)))
            if not 0 <= f.first_near_tie <= d:
                unflagged.append((seed, kind, f.status, want.status, d, f.first_near_tie))
    assert not unflagged, unflagged
    assert not flagged_continuous, flagged_continuous


def test_near_tie_stop_is_resumable(core):
    """opts.near_tie_action = STOP ends the run BEFORE the ambiguous pivot with DZG_NEAR_TIE, the
    state being that of the last executed pivot (a prefix of the oracle's log); running again
    takes the decision as FAST sees it and counts it."""
    from tests.lp_families import log3, make_lp

    a, b, c = make_lp(7001, 1, 30, 40)  # small integers: ties from the first pivot on
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=2000)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, near_tie_action=core.NEAR_TIE_STOP,
                     max_iter=2000) as s:
        assert s.run(0) == "near_tie"
        r = s.result()
        assert r.status == "near_tie" and r.near_ties == 0
        assert log3(r.pivots) == log3(want.pivots)[:r.iterations]
        stops = 0
        while s.run(0) == "near_tie":
            stops += 1
            assert stops < 2000
        r2 = s.result()
    assert r2.near_ties >= 1 and r2.first_near_tie == r.iterations
    assert r2.iterations > r.iterations or r2.status != "near_tie"
    assert len(r2.margins) == len(r2.pivots) and r2.min_margin <= r2.margins.min()


def test_state_drift_is_measured_at_a_refactorisation(core):
    """the near-tie tolerance followed the INVERSE's consistency only (64 x
    max_pivot_error, which a fresh inverse resets), while the carried x, xbar, z keep the rounding of
    every pivot so far.  At a refactorisation they are now recomputed from the fresh inverse
    (csrc/k_drift.hip: B^-1 b, B^-1 xbar0, N^T B^-T c_B - c_N), the largest relative difference is
    reported as state_drift and tau = max(tie_tol, 64 max_pivot_error, 4 state_drift).  512 x 1024
    against the oracle's 7 692 pivots with a refactorisation every 2 000: the pivots are the
    oracle's (COUNT mode: a flag changes no decision), the drift is measured, tiny and not zero,
    and every pivot after the last refactorisation whose margin lies inside 4 x drift is booked."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_2001_512x1024.npz"))
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=int(fx["m"]), n_struct=int(fx["n_struct"]))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    plain = core.solve(lp, numerics=core.FAST, poll_interval=50)
    assert plain.state_drift == 0.0 and plain.refactors == 0          # (never measured: no refactorisation)
    r = core.solve(lp, numerics=core.FAST, poll_interval=50, refactor_interval=2000)
    assert r.status == "optimal" and r.iterations == int(fx["iterations"]) and r.refactors == 3
    assert np.array_equal(np.array([p[1] for p in r.pivots]), fx["entering"])
    assert np.array_equal(np.array([p[2] for p in r.pivots]), fx["leaving"])
    assert 0.0 < r.state_drift < 1e-7, r.state_drift
    tau = max(1e-11, 4.0 * r.state_drift)
    late = r.margins[6000:]                                             # (after the last refactorisation)
    assert r.near_ties >= int((late <= tau).sum())
    # the oracle's mu against FAST's: the drift the tolerance now knows about is of that order
    mu_gap = max(abs(p[3] - w) / max(1.0, abs(w)) for p, w in zip(r.pivots, fx["mu"]))
    assert mu_gap < 1e-6
    print(f"state_drift {r.state_drift:.3e}, near_ties {r.near_ties}, first {r.first_near_tie}, "
          f"FAST-vs-oracle mu gap {mu_gap:.3e}, min margin {r.min_margin:.3e}")


def test_row_wise_pricing_alone_follows_the_oracle_log(core, monkeypatch):
    """VERDICT r3 (weak 1d): the evidence for the row-wise pricing pass against the ORACLE was indirect
    (whole-log tests run it while k < rows_T only).  Here the rule is forced to "always"
    (DZG_PRICE_ROWS_T above m): every one of the 7 692 pivots of the 512 x 1024 oracle log is priced
    over the rows v does not zero -- k up to 512 = m, the full 32 row groups -- and FAST still takes
    the oracle's pivots one by one, nothing flagged; the result says which pass ran."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_2001_512x1024.npz"))
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=int(fx["m"]), n_struct=int(fx["n_struct"]))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    monkeypatch.setenv("DZG_PRICE_ROWS_T", "1000000")
    rows = core.solve(lp, numerics=core.FAST, poll_interval=50)
    monkeypatch.setenv("DZG_PRICE_ROWS", "0")
    cols = core.solve(lp, numerics=core.FAST, poll_interval=50)
    monkeypatch.delenv("DZG_PRICE_ROWS", raising=False)
    monkeypatch.delenv("DZG_PRICE_ROWS_T", raising=False)
    for r, mask, copy in ((rows, 1, 1), (cols, 2, 0)):
        assert r.status == "optimal" and r.iterations == int(fx["iterations"]) and r.near_ties == 0
        assert np.array_equal(np.array([p[1] for p in r.pivots]), fx["entering"])
        assert np.array_equal(np.array([p[2] for p in r.pivots]), fx["leaving"])
        # (mu shrinks to 3e-4 over the solve while the carried state keeps its rounding: 1e-9 of the
        # FIRST pivots' mu, not of the last ones')
        assert np.allclose([p[3] for p in r.pivots], fx["mu"], rtol=1e-6, atol=1e-9)
        assert (r.price_pass_used, r.price_rows_copy) == (mask, copy)


def test_continuous_lp_is_not_flagged(core):
    """BASELINE config 2's family at 512 x 1024: thousands of pivots on continuous data, the
    oracle's log taken pivot for pivot, no near tie met -- AUTO keeps FAST's answer."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_2001_512x1024.npz"))
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=int(fx["m"]), n_struct=int(fx["n_struct"]))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    r = core.core_solve(lp, numerics=core.AUTO, log_cap=1 << 14)
    assert r.numerics == "fast" and r.status == "optimal" and r.near_ties == 0
    assert r.iterations == int(fx["iterations"]) and r.min_margin > 1e-11
    assert np.array_equal(np.array([p[1] for p in r.pivots]), fx["entering"])


# ------------------------------------------------------------------ k_price_tree
def test_tree_pricing_does_not_depend_on_its_launch_shape(core):
    """k_price_tree's sums depend only on m: the same column priced among 5, 700, 3000 or all
    17000 columns -- every pass width from 1 to 16 columns per wave (32 ... 2 tiles in flight), one
    pass and two -- gives the same bits, which is what makes a column-sharded solve take the pivots
    of the unsharded one."""
    rng = np.random.default_rng(5)
    m, ns = 1000, 17000
    a = rng.uniform(-1, 1, (m, ns))
    v = rng.uniform(-1, 1, m)
    full = core.neg_t_dot(a, np.arange(ns), v, kernel=core.PRICE_TREE)          # 17 per wave: 2 x 9
    assert np.allclose(full, -(a.T @ v), rtol=0, atol=1e-12)
    counts = [5, 700] + [1024 * w - 300 for w in range(2, 17)] + [1024 * 16, 16000]
    for count in counts:
        cols = rng.choice(ns, count, replace=False)
        part = core.neg_t_dot(a, cols, v, kernel=core.PRICE_TREE)
        assert_bit_equal(part, full[cols], f"{count} columns")
    # unit (slack) columns and a ragged row count
    m2 = 333
    a2 = rng.uniform(-1, 1, (m2, 40))
    v2 = rng.uniform(-1, 1, m2)
    cols2 = np.array([3, -1, 17, -m2, 39, -5])
    got = core.neg_t_dot(a2, cols2, v2, kernel=core.PRICE_TREE)
    want = np.array([-(a2[:, j] @ v2) if j >= 0 else -v2[-1 - j] for j in cols2])
    assert np.allclose(got, want, rtol=0, atol=1e-13)


# ------------------------------------------------------------------ refactorisation policy
def test_refactor_workspace_is_reserved_on_first_need(core):
    """refactor_interval = 0 reserves nothing at creation, yet the solver can still rebuild its
    inverse (dzg_solver_refactor reserves the workspace lazily -- what the pivot-consistency
    monitor relies on to recover instead of stopping with DZG_SINGULAR); the rebuilt inverse
    carries the solve on along the oracle's pivots, and the health monitor restarts with it:
    one refactorisation stays one, it is not repeated batch after batch."""
    a, b, c = core.gen_dense_lp(seed=91, m=96, n_struct=200)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, poll_interval=8) as s:
        assert s.run(150) == "iter_limit"
        before = s.result(log=False)
        assert before.refactors == 0 and before.max_pivot_error > 0.0
        s.refactor()
        assert s.run(40) == "iter_limit"
        mid = s.result(log=False)
        assert mid.refactors == 1
        assert s.run(0) == want.status == "optimal"
        res = s.result()
    assert res.refactors == 1                       # no refactorisation per batch afterwards
    assert res.max_pivot_error >= before.max_pivot_error   # the reported maximum is lifetime
    assert _log(res) == _log(want)


def test_sparse_basis_with_dense_rows_reaches_the_highs_optimum(core):
    """CSC input whose first two constraint rows are dense (every column has an entry there, as a
    budget row of a user model would): once more than 1024 structurals are basic, the leaving-slack
    BTRAN of those rows combines more rows of X than one LDS chunk holds (SP_LCAP) and the
    basic-slack FTRAN walks lists of that length.  The whole FAST solve on the sparse-basis path
    must reach the optimum an independent solver (scipy / HiGHS) finds, to 1e-9 relative."""
    import scipy.sparse as sp
    from scipy.optimize import linprog

    rng = np.random.default_rng(77)
    m, ns, per_col = 2200, 4000, 5
    rows = np.concatenate([np.sort(rng.choice(np.arange(2, m), per_col, replace=False)) for _ in range(ns)])
    cols = np.repeat(np.arange(ns), per_col)
    vals = rng.uniform(-1, 1, ns * per_col)
    a = sp.csc_matrix((vals, (rows, cols)), shape=(m, ns)).tolil()
    a[0, :] = rng.uniform(0.1, 1.0, ns)          # two dense rows
    a[1, :] = rng.uniform(-1.0, 1.0, ns)
    a = sp.csc_matrix(a)
    a.sort_indices()
    x0, y0 = rng.uniform(0, 1, ns), rng.uniform(0, 1, m)
    b = a @ x0 + rng.uniform(0, 1, m)            # x0 is feasible
    c = a.T @ y0 - rng.uniform(0, 1, ns)         # y0 is dual feasible: bounded
    lp = core.CoreLP.from_csc(m, a.indptr, a.indices, a.data, b, c)
    res = core.solve(lp, log=False, numerics=core.FAST, poll_interval=128)
    assert res.status == "optimal" and res.dense_columns > 1024
    assert res.max_pivot_error < 1e-8
    ref = linprog(-c, A_ub=a, b_ub=b, bounds=(0, None), method="highs")
    assert ref.status == 0
    assert abs(res.objective - (-ref.fun)) <= 1e-9 * max(1.0, abs(ref.fun))


def test_sparse_basis_path_leaves_the_oracle_path_only_where_it_flagged_a_near_tie(core):
    """The three LP families handed over as CSC (FAST runs on the sparse-basis path, k_sparse.hip:
    rows and columns of X appended, deleted and recycled, per-row lists of basic entries): 120 small
    LPs; continuous data follows the oracle with nothing flagged, any divergence on integer / 0-1
    data was flagged at or before the differing pivot, and AUTO equals the oracle throughout."""
    import scipy.sparse as sp

    from tests.lp_families import log3, make_lp

    unflagged, bad_auto, bad_continuous = [], [], []
    for seed in range(7300, 7420):
        kind = seed % 3
        a, b, c = make_lp(seed, kind, 2, 50)
        want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=5000)
        acsc = sp.csc_matrix(a)
        acsc.eliminate_zeros()
        acsc.sort_indices()
        lp = core.CoreLP.from_csc(a.shape[0], acsc.indptr, acsc.indices, acsc.data, b, c)
        f = core.solve(lp, numerics=core.FAST, max_iter=5000, poll_interval=8)
        got, wlog = log3(f.pivots), log3(want.pivots)
        if kind == 0 and (f.near_ties or got != wlog or f.status != want.status):
            bad_continuous.append((seed, f.near_ties, f.status, want.status))
        if got != wlog or f.status != want.status:
            d = next((i for i, (p, q) in enumerate(zip(got, wlog)) if p != q),
                     min(len(got), len(wlog)))
            if not 0 <= f.first_near_tie <= d:
                unflagged.append((seed, kind, f.status, want.status, d, f.first_near_tie))
        r = core.core_solve(lp, numerics=core.AUTO, max_iter=5000, auto_strict_rows=1, log_cap=5000)
        if r.status != want.status or log3(r.pivots) != wlog:
            bad_auto.append((seed, kind, r.status, want.status, r.numerics))
    assert not unflagged, unflagged
    assert not bad_continuous, bad_continuous
    assert not bad_auto, bad_auto


def _live_lists_bad(s):
    import ctypes as C

    from dantzig_amd import _ffi

    entries = C.c_int64(0)
    bad = _ffi.lib().dzg_debug_live_lists(s._h, C.byref(entries))
    assert bad >= 0, _ffi.lib().dzg_last_error()
    return bad, entries.value


def test_live_entry_pricing_keeps_its_lists_and_agrees_with_the_full_pass(core, monkeypatch):
    """The sparse-basis path prices over the LIVE entries of a column only (rows whose slack is
    nonbasic, csrc/k_price_kernels.h k_price_csc_rl): the lists are kept by k_sp_btran (a leaving
    slack's row joins) and k_sp_pivot (an entering slack's row goes), rebuilt at a refactorisation.
    After every few pivots the lists equal their definition (host recomputation through the test
    hook); the solve is the one the full pass (DZG_SP_PRICE_FULL=1: every stored entry) takes --
    same status and pivot log wherever neither flagged a near tie, same objective to 1e-9."""
    import scipy.sparse as sp

    from tests.lp_families import log3, make_lp

    cases = []
    for seed in range(8800, 8830):
        a, b, c = make_lp(seed, seed % 3, 2, 60)
        acsc = sp.csc_matrix(a)
        acsc.eliminate_zeros()
        acsc.sort_indices()
        cases.append((seed, core.CoreLP.from_csc(a.shape[0], acsc.indptr, acsc.indices, acsc.data, b, c), 0))
    for seed, m, ns, per_col, interval in [(8901, 300, 800, 5, 0), (8902, 500, 1200, 8, 0),
                                           (8903, 256, 700, 6, 41)]:
        cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
        cases.append((seed, core.CoreLP.from_csc(m, cp, ri, val, b, c), interval))
    most = 0
    for seed, lp, interval in cases:
        monkeypatch.delenv("DZG_SP_PRICE_FULL", raising=False)
        with core.Solver(lp, numerics=core.FAST, poll_interval=8, refactor_interval=interval) as s:
            chunks = 0
            while s.run(29) == "iter_limit" and chunks < 400:
                bad, entries = _live_lists_bad(s)
                assert bad == 0, (seed, chunks, bad)
                most = max(most, entries)
                chunks += 1
            live = s.result()
            bad, _ = _live_lists_bad(s)
            # (a dual step that ends at its ratio test has listed the leaving slack's row already:
            # ctl->rl_listed says so and the hook counts that row as listed -- consistent in every
            # final state, resumable or not)
            assert bad == 0, (seed, "end", bad, live.status)
        monkeypatch.setenv("DZG_SP_PRICE_FULL", "1")
        with core.Solver(lp, numerics=core.FAST, poll_interval=8, refactor_interval=interval) as s:
            chunks = 0
            while s.run(29) == "iter_limit" and chunks < 400:
                chunks += 1
            with pytest.raises(AssertionError):
                _live_lists_bad(s)  # (no lists are kept in this mode: the hook says so)
            full = s.result()
        if live.near_ties == 0 and full.near_ties == 0:
            assert live.status == full.status, seed
            assert log3(live.pivots) == log3(full.pivots), seed
        if live.status == full.status == "optimal":
            assert abs(live.objective - full.objective) <= 1e-9 * max(1.0, abs(full.objective)), seed
    assert most > 0  # (some row did join R somewhere)


def test_live_lists_survive_a_near_tie_stop_and_resume(core, monkeypatch):
    """k_sp_btran lists the leaving slack's row BEFORE the dual ratio test, which
    can stop the run with DZG_NEAR_TIE (resumable).  The resumed iteration must not list the row a
    second time (dz would count it twice from then on).  Integer / 0-1 LPs on the sparse-basis path
    in STOP mode, resumed at every stop: the lists equal their definition at every stop and at the
    end, and the solve is, bit for bit, the one COUNT mode takes (a resume decides the pivot as FAST
    sees it, which is all COUNT does)."""
    import scipy.sparse as sp

    from tests.lp_families import make_lp

    monkeypatch.delenv("DZG_SP_PRICE_FULL", raising=False)
    stops_seen = dual_stops = compared = 0
    for seed in list(range(8840, 8870)) + [8905]:
        if seed == 8905:
            rng = np.random.default_rng(seed)
            a = (rng.uniform(size=(220, 500)) < 0.04) * rng.integers(-3, 4, (220, 500)).astype(np.float64)
            b = rng.integers(-2, 9, 220).astype(np.float64)
            c = rng.integers(-4, 5, 500).astype(np.float64)
        else:
            a, b, c = make_lp(seed, 1 + seed % 2, 2, 60)
        acsc = sp.csc_matrix(a)
        acsc.eliminate_zeros()
        acsc.sort_indices()
        lp = core.CoreLP.from_csc(a.shape[0], acsc.indptr, acsc.indices, acsc.data, b, c)
        want = core.solve(lp, numerics=core.FAST, poll_interval=8, max_iter=4000)
        with core.Solver(lp, numerics=core.FAST, poll_interval=8, max_iter=4000,
                         near_tie_action=core.NEAR_TIE_STOP) as s:
            status, stops = s.run(0), 0
            while status == "near_tie":
                bad, _ = _live_lists_bad(s)
                assert bad == 0, (seed, "stop", stops, bad)
                dual_stops += int(_ffi_ctl_rl_listed(s) >= 0)
                stops += 1
                assert stops < 5000
                status = s.run(0)
            got = s.result()
            bad, _ = _live_lists_bad(s)
            assert bad == 0, (seed, "end", bad, got.status)
        stops_seen += stops
        # (the health monitor acts between BATCHES, and a stop cuts the batches differently: on these
        # integer LPs -- exact zeros, where its relative measure has no scale -- it can refactorise or
        # give up in one run and not in the other.  Wherever it stayed out of both, the solves agree
        # bit for bit.)
        if want.refactors == got.refactors == 0 and "singular" not in (want.status, got.status):
            compared += 1
            assert (got.status, got.iterations, got.pivots) == (want.status, want.iterations, want.pivots), seed
            assert np.array_equal(got.x, want.x) and np.array_equal(got.z, want.z), seed
            # (one pivot can stop twice -- at status() and again at its ratio test -- and a terminal
            # verdict inside the tolerance is a stop too: at least as many stops as booked near ties)
            assert got.near_ties == want.near_ties <= stops, (seed, got.near_ties, want.near_ties, stops)
    assert stops_seen > 20 and dual_stops > 0  # (some stop fell between BTRAN and the pivot)
    assert compared >= 20


def _ffi_ctl_rl_listed(s):
    """ctl->rl_listed: the constraint row k_sp_btran has listed ahead of a pivot that has not been
    executed (the run stopped between BTRAN and the pivot), -1 if none."""
    import ctypes as C

    from dantzig_amd import _ffi

    fn = _ffi.lib().dzg_debug_rl_listed
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int64
    return int(fn(s._h))


def test_auto_strict_resolve_with_a_zero_filled_max_iter(core):
    """dzg_core_solve's STRICT re-solve loop compared the pivot count with the
    CALLER's max_iter; a C host that zero-fills dzg_opts (0 = default) got DZG_ITER_LIMIT after the
    first 256-pivot chunk.  An integer LP whose STRICT solve needs more than 256 pivots, max_iter = 0:
    the outcome is the one max_iter = 10 000 000 gives."""
    from tests.lp_families import log3, make_lp

    found = None
    for seed in range(9100, 9140):
        a, b, c = make_lp(seed, 1, 200, 260)
        lp = core.CoreLP.from_inequality_form(a, b, c)
        r = core.core_solve(lp, numerics=core.AUTO, max_iter=10_000_000, log_cap=1 << 14)
        if r.numerics == "strict" and r.iterations > 300 and r.status != "iter_limit":
            found = (seed, lp, r)
            break
    assert found, "no integer LP with a STRICT re-solve of more than 300 pivots among the seeds"
    seed, lp, want = found
    got = core.core_solve(lp, numerics=core.AUTO, max_iter=0, log_cap=1 << 14)
    assert (got.status, got.numerics, got.iterations) == (want.status, "strict", want.iterations), seed
    assert log3(got.pivots) == log3(want.pivots)


def test_row_wise_pricing_agrees_with_the_column_pass(core, monkeypatch):
    """FAST with AUTO pricing prices an iteration ROW-wise while the compact width k is below
    rows_T = 0.9 m n_s / (n_s + m) (csrc/k_price_kernels.h k_price_rows: v is zero outside the k
    rows whose slack is nonbasic and the leaving slack's row, so k + 1 rows of a row-major copy
    replace m rows of every column), column-wise beyond.  The same solve as with the column pass
    alone (DZG_PRICE_ROWS=0): status, pivot log wherever neither run flagged a near tie, objective
    to 1e-9, and the two computations of the pivot element (FTRAN's dx_p, pricing's -dz_r) agree
    in both -- on the three LP families (k rises and falls across rows_T) and on a dense random LP
    whose whole solve crosses it; forcing rows_T above m prices every iteration row-wise."""
    from tests.lp_families import log3, make_lp

    lps = []
    for seed in range(9300, 9340):
        a, b, c = make_lp(seed, seed % 3, 2, 70)
        lps.append((seed, core.CoreLP.from_inequality_form(a, b, c)))
    a, b, c = core.gen_dense_lp(seed=9399, m=192, n_struct=448)
    lps.append((9399, core.CoreLP.from_inequality_form(a, b, c)))
    compared = 0
    for seed, lp in lps:
        runs = {}
        for mode in ("rows", "columns", "rows_always"):
            monkeypatch.delenv("DZG_PRICE_ROWS", raising=False)
            monkeypatch.delenv("DZG_PRICE_ROWS_T", raising=False)
            if mode == "columns":
                monkeypatch.setenv("DZG_PRICE_ROWS", "0")
            if mode == "rows_always":
                monkeypatch.setenv("DZG_PRICE_ROWS_T", "1000000")
            runs[mode] = core.solve(lp, numerics=core.FAST, max_iter=20000, poll_interval=16)
        want = runs["columns"]
        for mode in ("rows", "rows_always"):
            got = runs[mode]
            if seed % 3 == 0: # continuous data (the integer / 0-1 families pivot on exact ties and
                # zeros, where the monitor's relative measure has no scale in either mode)
                assert got.max_pivot_error < 1e-8, (seed, mode, got.max_pivot_error)
            if got.near_ties == 0 and want.near_ties == 0:
                assert got.status == want.status, (seed, mode)
                assert log3(got.pivots) == log3(want.pivots), (seed, mode)
                compared += 1
            if got.status == want.status == "optimal":
                assert abs(got.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective)), (seed, mode)
    assert compared >= 30
    # several column tiles and up to a dozen row groups: 512 x 2048, the first 3000 pivots
    a, b, c = core.gen_dense_lp(seed=9398, m=512, n_struct=2048)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    monkeypatch.delenv("DZG_PRICE_ROWS", raising=False)
    monkeypatch.delenv("DZG_PRICE_ROWS_T", raising=False)
    rows = core.solve(lp, numerics=core.FAST, max_iter=3000, poll_interval=50)
    monkeypatch.setenv("DZG_PRICE_ROWS", "0")
    cols = core.solve(lp, numerics=core.FAST, max_iter=3000, poll_interval=50)
    assert rows.dense_columns > 150 and rows.max_pivot_error < 1e-9
    assert log3(rows.pivots) == log3(cols.pivots)
    # more nonbasic positions than the finishing launch has threads (65 536): 40 x 70 000
    a, b, c = core.gen_dense_lp(seed=9397, m=40, n_struct=70000)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    monkeypatch.delenv("DZG_PRICE_ROWS", raising=False)
    monkeypatch.setenv("DZG_PRICE_ROWS_T", "1000000")
    rows = core.solve(lp, numerics=core.FAST, max_iter=400, poll_interval=50)
    monkeypatch.delenv("DZG_PRICE_ROWS_T", raising=False)
    monkeypatch.setenv("DZG_PRICE_ROWS", "0")
    cols = core.solve(lp, numerics=core.FAST, max_iter=400, poll_interval=50)
    assert rows.status == cols.status and rows.max_pivot_error < 1e-9
    if rows.near_ties == 0 and cols.near_ties == 0:
        assert log3(rows.pivots) == log3(cols.pivots)
    monkeypatch.delenv("DZG_PRICE_ROWS", raising=False)
    monkeypatch.delenv("DZG_PRICE_ROWS_T", raising=False)


# ------------------------------------------------------------------ three launches == seven launches
def _same_solution(r, w):
    return (r.status == w.status and r.iterations == w.iterations and r.pivots == w.pivots
            and r.near_ties == w.near_ties and r.first_near_tie == w.first_near_tie
            and all(_same_bits(getattr(r, f), getattr(w, f)) for f in ("x", "xbar", "z", "zbar"))
            and _same_bits(r.margins, w.margins) and r.max_pivot_error == w.max_pivot_error)


def test_chain_and_seven_launches_are_the_same_solve(core):
    """FAST on one GPU runs an iteration as three launches with device-wide barriers inside
    (csrc/k_chain.hip); column-sharded solvers (and opts.seven_launches) run it as seven.  Same row,
    dot-product and book-keeping functions: pivot log, mu, margins, monitor and the four vectors are
    bit-identical -- on continuous data and on the integer / 0-1 families, whose pivots append and
    delete compact columns (slack leaves, slack enters, both at once) and end unbounded,
    infeasible, optimal or at the iteration cap."""
    from tests.lp_families import make_lp

    bad, statuses = [], {}
    for seed in range(9100, 9190):
        a, b, c = make_lp(seed, seed % 3, 2, 90)
        lp = core.CoreLP.from_inequality_form(a, b, c)
        r3 = core.solve(lp, numerics=core.FAST, max_iter=3000, poll_interval=16)
        r7 = core.solve(lp, numerics=core.FAST, max_iter=3000, poll_interval=16, seven_launches=1)
        statuses[r3.status] = statuses.get(r3.status, 0) + 1
        if not _same_solution(r3, r7):
            bad.append((seed, a.shape, r3.status, r7.status, r3.iterations, r7.iterations))
    assert not bad, bad
    assert len(statuses) >= 3, statuses


def test_kernel_timing_can_sample_every_nth_pivot(core):
    """opts.profile bits 16..23: a sampling stride.  Event pairs cost idle GPU time between short
    kernels, so bench.py times every 8th pricing pass only: the stamped launches are the last of
    every 8 iterations of a batch, the timed classes the low 16 bits', and the solve is the solve
    (timing changes nothing on the device)."""
    from dantzig_amd import _ffi

    a, b, c = core.gen_dense_lp(seed=4242, m=96, n_struct=200)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    plain = core.solve(lp, numerics=core.FAST, max_iter=160, poll_interval=40)
    every = core.solve(lp, numerics=core.FAST, max_iter=160, poll_interval=40, profile=1 << _ffi.K_PRICE)
    some = core.solve(lp, numerics=core.FAST, max_iter=160, poll_interval=40,
                      profile=(1 << _ffi.K_PRICE) | (8 << 16))
    assert plain.iterations == every.iterations == some.iterations == 160
    assert _same_solution(plain, every) and _same_solution(plain, some)
    assert every.kernel_launches["price"] == 160
    assert some.kernel_launches["price"] == 4 * 5  # slots 7, 15, 23, 31, 39 of each batch of 40
    assert some.kernel_launches["update"] == 0 and some.kernel_ms["price"] > 0.0
    avg_all = every.kernel_ms["price"] / 160
    avg_some = some.kernel_ms["price"] / 20
    assert 0.2 * avg_all < avg_some < 5 * avg_all


def test_chain_follows_the_oracle_and_survives_stops_and_refactorisations(core):
    """The chain under everything a run can be cut by: budgets that end between its launches'
    iterations, near-tie stops (resumed), a refactorisation every 50 pivots -- against the
    oracle's pivot log on a continuous LP of 300 rows (eta flushes every 64 pivots, several
    thousand pivots)."""
    from tests.lp_families import log3

    m, ns = 300, 600
    a, b, c = core.gen_dense_lp(seed=9201, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(np.array(a), b, c), max_iter=200000)
    wlog = log3(want.pivots)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, poll_interval=7, refactor_interval=50,
                     near_tie_action=core.NEAR_TIE_STOP) as s:
        status, runs = "iter_limit", 0
        while status in ("iter_limit", "near_tie"):
            status = s.run(37)
            runs += 1
            assert runs < 100000
        r = s.result()
    assert status == want.status == r.status
    assert log3(r.pivots) == wlog
    assert r.refactors >= len(wlog) // 100
    assert abs(r.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))


def test_chain_hands_over_to_seven_launches_when_the_inverse_outgrows_lds(core, monkeypatch):
    """The chain keeps the gathered entering column in LDS (16 384 doubles); a batch that could
    outgrow it runs as seven launches instead.  With the cap lowered to 24 columns the two forms
    alternate within one solve (k rises and falls around the cap): still the seven-launch solve
    bit for bit, and the oracle's pivot log."""
    from tests.lp_families import log3

    a, b, c = core.gen_dense_lp(seed=9301, m=120, n_struct=260)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    r7 = core.solve(lp, numerics=core.FAST, poll_interval=5, seven_launches=1)
    monkeypatch.setenv("DZG_CHAIN_KCAP", "24")
    rmix = core.solve(lp, numerics=core.FAST, poll_interval=5)
    monkeypatch.delenv("DZG_CHAIN_KCAP")
    r3 = core.solve(lp, numerics=core.FAST, poll_interval=5)
    assert _same_solution(rmix, r7) and _same_solution(r3, r7)
    assert r7.dense_columns > 24  # the cap was crossed
    want = ora.simplex_solve(ora.stdform_from_dense(np.array(a), b, c), max_iter=100000)
    assert log3(r3.pivots) == log3(want.pivots) and r3.status == want.status


def test_chain_on_a_grid_smaller_than_the_eta_file(core, monkeypatch):
    """A device (or partition) with fewer CUs than the eta file has rows -- 64: beta_t = W_t . a_j is
    computed by workgroup t, so a grid of 8 or 24 workgroups must take several rows each (ADVICE r2:
    a grid below 64 left beta[grid..neta) stale).  DZG_CHAIN_GRID forces the grid; the solve must
    stay the seven-launch solve bit for bit, and the oracle's pivot log."""
    from tests.lp_families import log3

    a, b, c = core.gen_dense_lp(seed=9401, m=200, n_struct=420)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    r7 = core.solve(lp, numerics=core.FAST, poll_interval=16, seven_launches=1)
    assert r7.iterations > 500  # the eta file fills (64 rows) and is flushed many times
    for grid in ("8", "24", "100"):
        monkeypatch.setenv("DZG_CHAIN_GRID", grid)
        r3 = core.solve(lp, numerics=core.FAST, poll_interval=16)
        assert _same_solution(r3, r7), grid
    monkeypatch.delenv("DZG_CHAIN_GRID")
    want = ora.simplex_solve(ora.stdform_from_dense(np.array(a), b, c), max_iter=100000)
    assert log3(r7.pivots) == log3(want.pivots) and r7.status == want.status == "optimal"


def test_chain_recovers_when_another_kernel_holds_compute_units(core):
    """The three-launch iteration needs all of its workgroups resident at once (device-wide barriers,
    csrc/chain_barrier.h).  With a co-tenant on the device -- here a kernel that keeps 8 CUs busy for
    3 s with 128 KB of LDS each, so no chain workgroup fits beside it -- a barrier gives up after
    2.4 s; it fails for EVERY workgroup alike, nothing of the iteration has been written yet, and the
    host carries on with the barrier-free seven launches (engine.hip, chain_recover) instead of
    returning DZG_E_DEVICE.  The solve must end optimal with the oracle's pivot log
    and, bit for bit, the undisturbed solve's numbers."""
    import ctypes as C

    from dantzig_amd import _ffi
    from tests.lp_families import log3

    a, b, c = core.gen_dense_lp(seed=9501, m=256, n_struct=512)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    calm = core.solve(lp, numerics=core.FAST, poll_interval=16)
    assert calm.status == "optimal" and calm.chain_fallbacks == 0
    with core.Solver(lp, numerics=core.FAST, poll_interval=16) as s:
        assert s.run(100) == "iter_limit"  # the chain is up and running
        _ffi.check(_ffi.lib().dzg_debug_hold_cus(C.c_int32(0), C.c_int32(8), C.c_double(3.0)), "hold")
        status = s.run(0)
        got = s.result()
    _ffi.check(_ffi.lib().dzg_debug_hold_wait(), "hold_wait")
    assert status == "optimal"
    assert got.chain_fallbacks >= 1, "the co-tenant did not get in the chain's way: the test tested nothing"
    assert _same_solution(got, calm)
    want = ora.simplex_solve(ora.stdform_from_dense(np.array(a), b, c), max_iter=100000)
    assert log3(got.pivots) == log3(want.pivots) and want.status == "optimal"


def test_sparse_four_launches_are_the_eight_launch_solve(core, monkeypatch):
    """the sparse-basis iteration runs as FOUR launches (k_sp_pre, pricing, k_sp_mid,
    k_sp_update: csrc/k_sparse.hip) instead of eight -- FTRAN's two halves, the ratio decision and
    BTRAN's row share a launch behind the chain's device-wide barrier, and so do the dual step's
    FTRAN and the pivot's books.  The phases are the eight kernels' bodies: same status, pivot log
    with mu, margins, near-tie record, monitor and vectors as DZG_SP_FUSED=0, bit for bit -- on the
    three CSC LP families (ties, every verdict), with budgeted runs and near-tie stops resumed, with
    a refactorisation every 41 pivots, and on a 2 000 x 5 000 LP whose columns span several staging
    chunks of nothing in particular."""
    import scipy.sparse as sp

    from tests.lp_families import make_lp

    cases = []
    for seed in range(9700, 9745):
        a, b, c = make_lp(seed, seed % 3, 2, 70)
        acsc = sp.csc_matrix(a)
        acsc.eliminate_zeros()
        acsc.sort_indices()
        cases.append((seed, core.CoreLP.from_csc(a.shape[0], acsc.indptr, acsc.indices, acsc.data, b, c),
                      dict(max_iter=4000)))
    for seed, m, ns, per_col, extra in [(9751, 300, 800, 5, {}), (9752, 500, 1200, 8, dict(refactor_interval=41)),
                                        (9753, 2000, 5000, 30, dict(max_iter=3000)),
                                        (9754, 400, 900, 300, dict(max_iter=1500))]:   # (columns of 300 entries: 2 chunks)
        cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
        cases.append((seed, core.CoreLP.from_csc(m, cp, ri, val, b, c), extra))
    verdicts = set()
    for seed, lp, extra in cases:
        runs = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("DZG_SP_FUSED", fused)
            with core.Solver(lp, numerics=core.FAST, poll_interval=8, near_tie_action=core.NEAR_TIE_STOP,
                             **extra) as s:
                cap, status, guard = extra.get("max_iter", 10 ** 9), s.run(37), 0
                while status in ("iter_limit", "near_tie") and guard < 20000:   # budgets and stops, resumed
                    if status == "iter_limit" and s.result(log=False).iterations >= cap:
                        break
                    status, guard = s.run(37), guard + 1
                runs[fused] = s.result()
        verdicts.add(runs["1"].status)
        assert _same_solution(runs["1"], runs["0"]), seed
        assert np.array_equal(runs["1"].z, runs["0"].z) and runs["1"].objective == runs["0"].objective, seed
    monkeypatch.delenv("DZG_SP_FUSED", raising=False)
    assert {"optimal", "unbounded", "infeasible"} <= verdicts


def test_chain_and_seven_launches_agree_over_a_whole_solve_of_config_2(core):
    """BASELINE config 2 (1024 x 2048, seed 1002) solved to optimality in both forms: some 21 600
    pivots, the compact inverse beyond 512 columns for most of them (the one-wave-per-row FTRAN), an
    eta flush every 64 pivots -- same pivot log, margins, monitor and vectors, bit for bit."""
    a, b, c = core.gen_dense_lp(seed=1002, m=1024, n_struct=2048)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    r3 = core.solve(lp, numerics=core.FAST, log_capacity=1 << 15)
    r7 = core.solve(lp, numerics=core.FAST, log_capacity=1 << 15, seven_launches=1)
    assert r3.status == "optimal" and r3.iterations > 20000 and r3.dense_columns > 512
    assert _same_solution(r3, r7)
    assert r3.objective == r7.objective

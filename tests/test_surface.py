"""The modelling surface (dantzig_amd.Variable / Minimize / Maximize / Solution).

CPU part: expression algebra, and the lowering of every Python known-answer problem of the
reference (tests/golden/reference_kats.json "python") fed to the oracle -- exact `==` like
the reference's own tests, which pins both the lowering and the oracle's arithmetic.
GPU part: the same problems through `.solve()`, i.e. through the HIP engine."""
import json
import os

import numpy as np
import pytest

import dantzig_amd as dz
from dantzig_amd import rust as rs
from dantzig_amd.model import AffExpr, Constraint, LinExpr
from oracle import oracle as ora

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as _f:
    PY_KATS = json.load(_f)["python"]


def make_var(spec):
    if spec == "nonneg":
        return dz.Variable.nonneg()
    if spec == "free":
        return dz.Variable.free()
    if spec == "nonpos":
        return dz.Var.np()
    return dz.Var(lb=spec[0], ub=spec[1])


def build_problem(k):
    ns = {name: make_var(spec) for name, spec in k["vars"].items()}
    env = dict(ns, sum=sum, zip=zip)
    objective = eval(k["objective"], {"__builtins__": {}}, env)
    cons = [eval(c, {"__builtins__": {}}, env) for c in k["constraints"]]
    cls = dz.Minimize if k["sense"] == "min" else dz.Maximize
    return ns, cls(objective).subject_to(cons)


def oracle_model(problem):
    """rs.lower() output -> the oracle's JSON-style model."""
    objective = problem._core_objective().to_rust_affexpr()
    arrays, order = rs.lower(objective, list(problem.yield_rust_inequalities()))
    cons = []
    for r in range(arrays["ncons"]):
        lo, hi = arrays["con_ptr"][r], arrays["con_ptr"][r + 1]
        cons.append({"terms": [[int(arrays["con_var"][e]), float(arrays["con_coef"][e])]
                               for e in range(lo, hi)], "b": float(arrays["con_b"][r])})
    model = {
        "vars": [{"lb": v.lb, "ub": v.ub} for v in order],
        "objective": {"terms": [[int(arrays["obj_var"][t]), float(arrays["obj_coef"][t])]
                                for t in range(arrays["obj_nterms"])],
                      "constant": float(arrays["obj_const"])},
        "constraints": cons,
    }
    return model, order


# ------------------------------------------------------------------ algebra (CPU)
def same(a, b):
    if isinstance(a, AffExpr) or isinstance(b, AffExpr):
        a, b = a.to_affexpr(), b.to_affexpr()
        return a.linexpr.map_ids_to_coefs() == b.linexpr.map_ids_to_coefs() \
            and a.constant == b.constant
    return a.to_linexpr().map_ids_to_coefs() == b.to_linexpr().map_ids_to_coefs()


def test_expression_types():
    x, y = dz.Variable.nonneg(), dz.Variable.free(name="y")
    assert isinstance(x + y, LinExpr) and isinstance(x - y, LinExpr) and isinstance(-x, LinExpr)
    assert isinstance(3 * x, LinExpr) and isinstance(x * 3.5, LinExpr)
    assert isinstance(x + 1, AffExpr) and isinstance(1 + x, AffExpr) and isinstance(1 - x, AffExpr)
    assert isinstance((x + y) + 2.0, AffExpr) and isinstance(2 * (x + 1), AffExpr)
    assert isinstance(x <= 1, Constraint) and isinstance(x == y, Constraint)
    assert isinstance(x + 1 >= y, Constraint)
    assert y.name == "y" and x.name is None and x.lb == 0.0 and x.ub is None
    assert y.lb is None and dz.Var.np().ub == 0.0
    assert len({x, y}) == 2 and x.id != y.id
    with pytest.raises(TypeError):
        x * y
    with pytest.raises(TypeError):
        (x + 1) * y
    with pytest.raises(TypeError):
        dz.Variable(lb=0.0)  # both bounds are required keywords
    with pytest.raises(TypeError):
        dz.Minimize(x).subject_to(3)


def test_linear_identities():
    x, y = dz.Variable.nonneg(), dz.Variable.nonneg()
    assert same(-x, -1.0 * x) and same(-x, x * -1.0)
    assert same(x + x, 2 * x) and same(x + y + x, y + 2 * x)
    assert same(x - y, -y + x) and same(-(x + y), -y - x)
    assert same(2 * x + 2 * y, (x + y) * 2) and same(x * 2 + y * 2, 2 * (x + y))
    assert same(2 * x - x, x.to_linexpr())
    assert list((x + y + x).map_ids_to_coefs()) == [x.id, y.id]  # first-seen order kept


def test_affine_identities():
    x, y = dz.Variable.free(), dz.Variable.free()
    e = x + y + 5
    assert same(x + 5.0, 5.0 + x) and same(2 * x + 2, (x + 1) * 2)
    assert same(e + e, 2 * x + 2 * y + 10) and same(e + e, 10.0 + 2 * x + 2 * y)
    assert same(e + e, e * 2) and same(-(x + y + 1), 0.0 - (x + y + 1))
    assert same(-(x + y + 1), -1 * (x + y + 1)) and same(x + y + 1, 0.0 + (x + y + 1))
    assert same(7 - e, -x - y + 2)


def test_constraint_lowering():
    x, y = dz.Variable.nonneg(), dz.Variable.nonneg()
    (le,) = (x + 2 * y <= 4).rust_inequalities()
    assert (le._linexpr.coefs, le._b) == ([1.0, 2.0], 4.0)
    (ge,) = (x >= y + 1).rust_inequalities()
    assert (ge._linexpr.coefs, ge._b) == ([-1.0, 1.0], -1.0)
    lo, hi = (y == 3).rust_inequalities()
    assert (lo._linexpr.coefs, lo._b, hi._linexpr.coefs, hi._b) == ([1.0], 3.0, [-1.0], -3.0)
    # a chained comparison keeps only its right half (Python `and` on a truthy Constraint)
    (kept,) = (-3.0 <= x <= 3.0).rust_inequalities()
    assert (kept._linexpr.coefs, kept._b) == ([1.0], 3.0)
    m = dz.Min(x).st(x <= 1).st([y <= 2, y >= 0])
    assert len(m.constraints) == 3 and m.sense == "minimize" and dz.Max(x).sense == "maximize"


# ------------------------------------------------------------------ Python KATs on the oracle (CPU)
@pytest.mark.parametrize("k", PY_KATS, ids=[k["name"] for k in PY_KATS])
def test_python_kat_lowering_on_oracle(k):
    ns, problem = build_problem(k)
    model, order = oracle_model(problem)
    res = ora.solve_model(model)
    exp = k["expect"]
    if "error" in exp:
        assert res.status == {"UnboundedError": "unbounded", "InfeasibleError": "infeasible"}[
            exp["error"]]
        return
    assert res.status == "optimal"
    core_obj = res.objective
    user_obj = -core_obj if k["sense"] == "min" else core_obj
    if "objective" in exp:
        assert user_obj == exp["objective"]  # exact, like the reference's tests
    value = {v.id: res.values[i] for i, v in enumerate(order)}
    for name, want in exp["values"].items():
        assert value.get(ns[name].id, 0.0) == want


def test_product_builder_matches_oracle_builder():
    """dzg_build_standard_form (host C++, product) == the oracle's Simplex::new restatement."""
    import ctypes as C

    from dantzig_amd import _ffi

    for k in PY_KATS:
        _, problem = build_problem(k)
        objective = problem._core_objective().to_rust_affexpr()
        arrays, order = rs.lower(objective, list(problem.yield_rust_inequalities()))
        model, _ = oracle_model(problem)
        want = ora.build_standard_form(model)
        md = rs._c_model(arrays)
        sf = _ffi.StdForm()
        assert _ffi.lib().dzg_build_standard_form(C.byref(md), C.byref(sf)) == 0
        m, n, nst = sf.m, sf.n, sf.n_struct
        assert (m, n) == (want.m, want.n)
        a = np.zeros((max(nst, 1), max(sf.lda, 1)))
        var_col, c = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1))
        basis, nonbasis = np.zeros(max(m, 1), np.int64), np.zeros(max(n - m, 1), np.int64)
        x, z = np.zeros(max(m, 1)), np.zeros(max(n - m, 1))
        pos, neg = np.zeros(len(order) + 1, np.int64), np.zeros(len(order) + 1, np.int64)
        p = _ffi.ptr
        sf.a, sf.var_col, sf.c, sf.basis, sf.nonbasis = p(a), p(var_col), p(c), p(basis), p(nonbasis)
        sf.x, sf.z, sf.pos_var, sf.neg_var = p(x), p(z), p(pos), p(neg)
        assert _ffi.lib().dzg_build_standard_form(C.byref(md), C.byref(sf)) == 0
        assert basis[:m].tolist() == want.basis.tolist()
        assert nonbasis[: n - m].tolist() == want.nonbasis.tolist()
        assert np.array_equal(x[:m], want.x) and np.array_equal(np.signbit(x[:m]), np.signbit(want.x))
        assert np.array_equal(z[: n - m], want.z) and np.array_equal(c[:n], want.c)
        assert pos[: len(order)].tolist() == want.pos_col.tolist()
        assert neg[: len(order)].tolist() == want.neg_col.tolist()
        dense = ora.csc_to_dense(m, n, want.col_ptr, want.row_idx, want.val)
        for v in range(n):
            col = a[var_col[v], :m] if var_col[v] >= 0 else np.eye(m)[-1 - var_col[v]]
            assert np.array_equal(col, dense[:, v])


def test_solve_without_gpu_fails_loudly():
    from dantzig_amd import _ffi

    if _ffi.lib().dzg_device_count() > 0:
        pytest.skip("a GPU is visible")
    x = dz.Variable.nonneg()
    with pytest.raises(_ffi.DantzigAmdError):
        dz.Minimize(x).solve()


# ------------------------------------------------------------------ Python KATs on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("k", PY_KATS, ids=[k["name"] for k in PY_KATS])
def test_python_kat_on_gpu(k):
    ns, problem = build_problem(k)
    exp = k["expect"]
    if "error" in exp:
        with pytest.raises(getattr(dz.exceptions, exp["error"])):
            problem.solve()
        return
    sol = problem.solve()
    if "objective" in exp:
        assert sol.objective_value == exp["objective"]
    for name, want in exp["values"].items():
        assert sol[ns[name]] == want
    assert sol[dz.Variable.nonneg()] == 0.0  # unknown variable -> 0.0 (src/pyobjs.rs:163-165)


# ------------------------------------------------------------------ larger user models on the GPU
@pytest.mark.gpu
def test_large_sparse_model_through_surface():
    """1500 bounded variables, 600 sparse rows: the standard form is 3600 x 6600 and mostly
    zero, so Level 2 hands the structural block over in CSC (never densified) and FAST numerics
    solves it; the optimum is checked against an independent solver (scipy / HiGHS)."""
    import scipy.sparse as sp
    from scipy.optimize import linprog

    rng = np.random.default_rng(3)
    K, M, per_row = 1500, 600, 6
    xs = [dz.Variable(lb=0.0, ub=10.0) for _ in range(K)]
    c = rng.uniform(0.1, 1.0, K)
    x0 = rng.uniform(0, 1, K)
    a = sp.lil_matrix((M, K))
    rows = []
    for r in range(M):
        idx = rng.choice(K, per_row, replace=False)
        coef = rng.uniform(0.1, 1.0, per_row)
        a[r, idx] = coef
        rows.append((idx, coef, float(coef @ x0[idx]) + rng.uniform(0.1, 1)))
    problem = dz.Maximize(sum(float(ci) * xi for ci, xi in zip(c, xs)))
    problem.subject_to([sum(float(cf) * xs[i] for i, cf in zip(idx, coef)) <= b
                        for idx, coef, b in rows])
    ref = linprog(-c, A_ub=a.tocsr(), b_ub=[b for _, _, b in rows], bounds=(0, 10), method="highs")
    sol = problem.solve()
    assert sol._solution.numerics == "fast" and sol._solution.shape == (3600, 6600)
    assert abs(sol.objective_value + ref.fun) <= 1e-9 * abs(ref.fun)
    got = np.array([sol[x] for x in xs])
    assert got.min() >= -1e-9 and got.max() <= 10 + 1e-9
    assert (a.tocsr() @ got - np.array([b for _, _, b in rows])).max() <= 1e-7


@pytest.mark.gpu
def test_degenerate_transportation_model_through_surface():
    """Integer data, heavy degeneracy (exact ties everywhere), 774 rows: AUTO starts in FAST, meets a
    tie within the first pivots and hands the model to STRICT (some 15 ms per pivot at this size)."""
    from scipy.optimize import linprog

    rng = np.random.default_rng(4)
    S, D = 24, 30
    supply = rng.integers(20, 60, S).astype(float)
    demand = rng.integers(5, 25, D).astype(float)
    cost = rng.integers(1, 20, (S, D)).astype(float)
    x = [[dz.Variable.nonneg() for _ in range(D)] for _ in range(S)]
    problem = dz.Minimize(sum(cost[i][j] * x[i][j] for i in range(S) for j in range(D)))
    problem.subject_to([sum(x[i][j] for j in range(D)) <= supply[i] for i in range(S)]
                       + [sum(x[i][j] for i in range(S)) >= demand[j] for j in range(D)])
    a = np.zeros((S + D, S * D))
    b = np.zeros(S + D)
    for i in range(S):
        a[i, i * D:(i + 1) * D] = 1
        b[i] = supply[i]
    for j in range(D):
        a[S + j, j::D] = -1
        b[S + j] = -demand[j]
    ref = linprog(cost.ravel(), A_ub=a, b_ub=b, bounds=(0, None), method="highs")
    sol = problem.solve()
    assert abs(sol.objective_value - ref.fun) <= 1e-9 * abs(ref.fun)


def _zero_one_model(seed, ncons=130, nvars=110):
    """A 0/1 packing model, 240 rows in standard form (> auto_strict_rows): degenerate vertices,
    0/0 ratios -- data on which an updated explicit inverse loses its footing."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(size=(ncons, nvars)) < 0.25
    b = rng.integers(0, 4, ncons)
    c = rng.integers(-1, 6, nvars)
    xs = [dz.Variable.nonneg() for _ in range(nvars)]
    objective = sum((float(c[j]) * xs[j] for j in range(1, nvars)), float(c[0]) * xs[0])
    cons = []
    for i in range(ncons):
        cols = np.nonzero(a[i])[0]
        if len(cols):
            cons.append(sum((xs[j] * 1.0 for j in cols[1:]), xs[cols[0]] * 1.0) <= float(b[i]))
    return dz.Maximize(objective).subject_to(cons)


def _outcome(seed, numerics):
    from dantzig_amd import _ffi

    rs.set_options(numerics=numerics)
    try:
        return ("optimal", _zero_one_model(seed).solve().objective_value)
    except Exception as exc:  # noqa: BLE001 -- the outcome IS the exception type and text
        return (type(exc).__name__, str(exc))
    finally:
        rs.set_options()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 4, 5, 11])
def test_auto_numerics_falls_back_to_strict_when_fast_gives_up(seed):
    """Above auto_strict_rows AUTO runs FAST; when FAST meets a near tie or loses its footing
    (DZG_SINGULAR / DZG_PANIC), Level 2 answers with the reference's own arithmetic (STRICT)
    instead of an error.  On these 0/1 models explicit FAST does lose it."""
    from dantzig_amd import _ffi

    fast = _outcome(seed, _ffi.FAST)
    assert fast[0] == "RuntimeError" and ("singular" in fast[1] or "panic" in fast[1])
    assert _outcome(seed, _ffi.AUTO) == _outcome(seed, _ffi.STRICT)

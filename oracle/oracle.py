"""ctypes front end of the CPU oracle (oracle/dzg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package dantzig_amd never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdzg_oracle.so")

STATUS = {0: "optimal", 1: "unbounded", 2: "infeasible", 3: "iter_limit", 5: "panic"}
PRIMAL, DUAL = 0, 1


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "dzg_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "dzg_oracle.h"))
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libdzg_oracle.so"])
    return _LIB_PATH


class _Simplex(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64),
        ("col_ptr", C.c_void_p), ("row_idx", C.c_void_p), ("val", C.c_void_p),
        ("c", C.c_void_p), ("constant", C.c_double),
        ("basis", C.c_void_p), ("nonbasis", C.c_void_p),
        ("x", C.c_void_p), ("xbar", C.c_void_p), ("z", C.c_void_p), ("zbar", C.c_void_p),
    ]


class _Pivot(C.Structure):
    _fields_ = [("kind", C.c_int32), ("entering", C.c_int64), ("leaving", C.c_int64),
                ("mu", C.c_double)]


class _Model(C.Structure):
    _fields_ = [
        ("nvars", C.c_int64), ("has_lb", C.c_void_p), ("has_ub", C.c_void_p),
        ("lb", C.c_void_p), ("ub", C.c_void_p),
        ("obj_nterms", C.c_int64), ("obj_var", C.c_void_p), ("obj_coef", C.c_void_p),
        ("obj_const", C.c_double),
        ("ncons", C.c_int64), ("con_ptr", C.c_void_p), ("con_var", C.c_void_p),
        ("con_coef", C.c_void_p), ("con_b", C.c_void_p),
    ]


class _StdForm(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("nnz", C.c_int64),
        ("col_ptr", C.POINTER(C.c_int64)), ("row_idx", C.POINTER(C.c_int64)),
        ("val", C.POINTER(C.c_double)), ("c", C.POINTER(C.c_double)),
        ("constant", C.c_double),
        ("basis", C.POINTER(C.c_int64)), ("nonbasis", C.POINTER(C.c_int64)),
        ("x", C.POINTER(C.c_double)), ("z", C.POINTER(C.c_double)),
        ("pos_col", C.POINTER(C.c_int64)), ("neg_col", C.POINTER(C.c_int64)),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ora_simplex_solve.restype = C.c_int
        _lib.ora_objective_value.restype = C.c_double
        _lib.ora_csc_from_dense.restype = C.c_int64
        _lib.ora_find_first_pivot.restype = C.c_int64
        _lib.ora_find_second_pivot.restype = C.c_int64
        _lib.ora_build_standard_form.restype = C.c_int
    return _lib


_BLOCKED_PATH = os.path.join(_HERE, "libdzg_oracle_blocked.so")
_blocked = None


def blocked_lib() -> C.CDLL:
    """The twin library: the same restatement with Matrix::factorize applied block by block on
    several cores (dzg_oracle_blocked.c: same operations per element in the same order, bit-equal
    factors -- tests/test_oracle_kats.py).  For pivot-log fixtures at benchmark sizes only; the
    literal library stays the arbiter and the CPU baseline."""
    global _blocked
    if _blocked is None:
        srcs = [os.path.join(_HERE, f) for f in ("dzg_oracle.c", "dzg_oracle_blocked.c", "dzg_oracle.h")]
        if (not os.path.exists(_BLOCKED_PATH)
                or os.path.getmtime(_BLOCKED_PATH) < max(os.path.getmtime(f) for f in srcs)):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libdzg_oracle_blocked.so"])
        _blocked = C.CDLL(_BLOCKED_PATH)
        _blocked.ora_simplex_solve.restype = C.c_int
        _blocked.ora_objective_value.restype = C.c_double
    return _blocked


def lu_factorize_blocked(a: np.ndarray):
    """ora_lu_factorize_blocked: must equal lu_factorize bit for bit."""
    a = _f64(a).copy()
    n = a.shape[0]
    p = np.zeros(max(n - 1, 1), dtype=np.int64)
    blocked_lib().ora_lu_factorize_blocked(_p(a), C.c_int64(n), _p(p))
    return a, p[: max(n - 1, 0)]


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


# --------------------------------------------------------------------------- linalg
def lu_factorize(a: np.ndarray):
    """Matrix::factorize. Returns (packed LU row-major n*n, p[n-1])."""
    a = _f64(a).copy()
    n = a.shape[0]
    p = np.zeros(max(n - 1, 1), dtype=np.int64)
    lib().ora_lu_factorize(_p(a), C.c_int64(n), _p(p))
    return a, p[: max(n - 1, 0)]


def lu_solve(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """lu_solve(a, b): factorize then solve."""
    a = _f64(a).copy()
    b = _f64(b).copy()
    lib().ora_lu_solve_full(_p(a), C.c_int64(a.shape[0]), _p(b))
    return b


def lu_solve_factored(lu: np.ndarray, p: np.ndarray, b: np.ndarray) -> np.ndarray:
    lu = _f64(lu)
    pp = _i64(np.concatenate([p, [0]]))
    b = _f64(b).copy()
    lib().ora_lu_solve(_p(lu), C.c_int64(lu.shape[0]), _p(pp), _p(b))
    return b


def matrix_t(a: np.ndarray) -> np.ndarray:
    a = _f64(a)
    out = np.empty((a.shape[1], a.shape[0]), dtype=np.float64)
    lib().ora_matrix_t(_p(a), C.c_int64(a.shape[0]), C.c_int64(a.shape[1]), _p(out))
    return out


def csc_from_dense(a: np.ndarray):
    a = _f64(a)
    nr, nc = a.shape
    col_ptr = np.zeros(nc + 1, dtype=np.int64)
    row_idx = np.zeros(max(nr * nc, 1), dtype=np.int64)
    val = np.zeros(max(nr * nc, 1), dtype=np.float64)
    nnz = lib().ora_csc_from_dense(_p(a), C.c_int64(nr), C.c_int64(nc), _p(col_ptr),
                                   _p(row_idx), _p(val))
    return col_ptr, row_idx[:nnz].copy(), val[:nnz].copy()


def csc_column(nrows, col_ptr, row_idx, val, j) -> np.ndarray:
    out = np.empty(nrows, dtype=np.float64)
    col_ptr, row_idx, val = _i64(col_ptr), _i64(row_idx), _f64(val)
    lib().ora_csc_column(C.c_int64(nrows), _p(col_ptr), _p(row_idx), _p(val), C.c_int64(j),
                         _p(out))
    return out


def csc_to_dense(nrows, ncols, col_ptr, row_idx, val) -> np.ndarray:
    out = np.empty((nrows, ncols), dtype=np.float64)
    col_ptr, row_idx, val = _i64(col_ptr), _i64(row_idx), _f64(val)
    lib().ora_csc_to_dense(C.c_int64(nrows), C.c_int64(ncols), _p(col_ptr), _p(row_idx),
                           _p(val), _p(out))
    return out


def neg_t_dot(col_ptr, row_idx, val, cols, v) -> np.ndarray:
    """collect_columns(cols).neg_t_dot(v)."""
    col_ptr, row_idx, val = _i64(col_ptr), _i64(row_idx), _f64(val)
    cols, v = _i64(cols), _f64(v)
    out = np.empty(len(cols), dtype=np.float64)
    lib().ora_csc_neg_t_dot(_p(col_ptr), _p(row_idx), _p(val), _p(cols),
                            C.c_int64(len(cols)), _p(v), _p(out))
    return out


def find_first_pivot(y, ybar) -> int:
    y, ybar = _f64(y), _f64(ybar)
    return int(lib().ora_find_first_pivot(_p(y), _p(ybar), C.c_int64(len(y))))


def find_second_pivot(mu, y, ybar, dy) -> int:
    y, ybar, dy = _f64(y), _f64(ybar), _f64(dy)
    return int(lib().ora_find_second_pivot(C.c_double(mu), _p(y), _p(ybar), _p(dy),
                                           C.c_int64(len(y))))


# --------------------------------------------------------------------------- simplex
@dataclass
class StdForm:
    """State of `Simplex` right after Simplex::new (general CSC over all n columns)."""
    m: int
    n: int
    col_ptr: np.ndarray
    row_idx: np.ndarray
    val: np.ndarray
    c: np.ndarray
    constant: float
    basis: np.ndarray
    nonbasis: np.ndarray
    x: np.ndarray
    z: np.ndarray
    pos_col: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    neg_col: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))


@dataclass
class SolveResult:
    status: str
    iterations: int
    objective: float
    basis: np.ndarray
    nonbasis: np.ndarray
    x: np.ndarray
    xbar: np.ndarray
    z: np.ndarray
    zbar: np.ndarray
    pivots: list  # (kind, entering, leaving, mu)
    values: np.ndarray | None = None


def stdform_from_dense(a: np.ndarray, b: np.ndarray, c: np.ndarray,
                       constant: float = 0.0) -> StdForm:
    """max c.x st A x <= b, x >= 0 entered directly at the core boundary
    (SURVEY 8(d) convention): columns 0..ns-1 structural, ns..ns+m-1 slacks."""
    a = _f64(a)
    m, ns = a.shape
    full = np.concatenate([a, np.eye(m)], axis=1)
    col_ptr, row_idx, val = csc_from_dense(full)
    cc = np.concatenate([_f64(c), np.zeros(m)])
    return StdForm(m=m, n=ns + m, col_ptr=col_ptr, row_idx=row_idx, val=val, c=cc,
                   constant=constant, basis=np.arange(ns, ns + m, dtype=np.int64),
                   nonbasis=np.arange(ns, dtype=np.int64), x=_f64(b).copy(),
                   z=-_f64(c))


def simplex_solve(sf: StdForm, max_iter: int = 1_000_000, log_cap: int | None = None,
                  blocked: bool = False) -> SolveResult:
    m, n = sf.m, sf.n
    q = n - m
    basis, nonbasis = _i64(sf.basis).copy(), _i64(sf.nonbasis).copy()
    x, z = _f64(sf.x).copy(), _f64(sf.z).copy()
    if len(x) == 0:
        x = np.zeros(0)
    xbar, zbar = np.ones(max(m, 1))[:m].copy(), np.ones(max(q, 1))[:q].copy()
    col_ptr, row_idx, val, c = _i64(sf.col_ptr), _i64(sf.row_idx), _f64(sf.val), _f64(sf.c)
    # keep 1-element backing stores alive for empty vectors
    keep = [np.zeros(1), np.zeros(1, dtype=np.int64)]
    st = _Simplex(m, n, _p(col_ptr), _p(row_idx if len(row_idx) else keep[1]),
                  _p(val if len(val) else keep[0]), _p(c if len(c) else keep[0]),
                  float(sf.constant),
                  _p(basis if m else keep[1]), _p(nonbasis if q else keep[1]),
                  _p(x if m else keep[0]), _p(xbar if m else keep[0]),
                  _p(z if q else keep[0]), _p(zbar if q else keep[0]))
    cap = int(log_cap if log_cap is not None else min(max_iter, 4_000_000))
    log = (_Pivot * max(cap, 1))()
    iters = C.c_int64(0)
    the_lib = blocked_lib() if blocked else lib()
    status = the_lib.ora_simplex_solve(C.byref(st), C.c_int64(max_iter), C.byref(iters), log,
                                       C.c_int64(cap))
    k = min(iters.value, cap)
    pivots = [(log[i].kind, log[i].entering, log[i].leaving, log[i].mu) for i in range(k)]
    obj = float(lib().ora_objective_value(C.byref(st))) if m >= 0 else float("nan")
    return SolveResult(STATUS[status], iters.value, obj, basis, nonbasis, x, xbar, z, zbar,
                       pivots)


# --------------------------------------------------------------------------- model
def build_standard_form(model: dict) -> StdForm:
    """Simplex::new over a JSON-style model (see tests/golden/reference_kats.json)."""
    vs = model["vars"]
    V = len(vs)
    has_lb = np.array([v.get("lb") is not None for v in vs] + [0], dtype=np.int32)
    has_ub = np.array([v.get("ub") is not None for v in vs] + [0], dtype=np.int32)
    lb = np.array([v["lb"] if v.get("lb") is not None else 0.0 for v in vs] + [0.0])
    ub = np.array([v["ub"] if v.get("ub") is not None else 0.0 for v in vs] + [0.0])
    ot = model["objective"]["terms"]
    obj_var = _i64([t[0] for t in ot] + [0])
    obj_coef = _f64([t[1] for t in ot] + [0.0])
    cons = model.get("constraints", [])
    con_ptr = np.zeros(len(cons) + 1, dtype=np.int64)
    cv, cc, cb = [], [], []
    for r, con in enumerate(cons):
        for t in con["terms"]:
            cv.append(t[0])
            cc.append(t[1])
        con_ptr[r + 1] = len(cv)
        cb.append(con["b"])
    con_var, con_coef, con_b = _i64(cv + [0]), _f64(cc + [0.0]), _f64(cb + [0.0])
    md = _Model(V, _p(has_lb), _p(has_ub), _p(lb), _p(ub), len(ot), _p(obj_var),
                _p(obj_coef), float(model["objective"].get("constant", 0.0)), len(cons),
                _p(con_ptr), _p(con_var), _p(con_coef), _p(con_b))
    out = _StdForm()
    rc = lib().ora_build_standard_form(C.byref(md), C.byref(out))
    assert rc == 0
    m, n, nnz = out.m, out.n, out.nnz

    def arr(ptr, count, dtype):
        if count <= 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype).copy()

    sf = StdForm(m=m, n=n, col_ptr=arr(out.col_ptr, n + 1, np.int64),
                 row_idx=arr(out.row_idx, nnz, np.int64), val=arr(out.val, nnz, np.float64),
                 c=arr(out.c, n, np.float64), constant=out.constant,
                 basis=arr(out.basis, m, np.int64), nonbasis=arr(out.nonbasis, n - m, np.int64),
                 x=arr(out.x, m, np.float64), z=arr(out.z, n - m, np.float64),
                 pos_col=arr(out.pos_col, V, np.int64), neg_col=arr(out.neg_col, V, np.int64))
    lib().ora_stdform_free(C.byref(out))
    return sf


def solution_values(sf: StdForm, res: SolveResult) -> np.ndarray:
    """Simplex::solution: x+ minus x- per user variable, 0.0 when nonbasic/unknown."""
    pos_of = {int(v): k for k, v in enumerate(res.basis)}
    out = np.zeros(len(sf.pos_col))
    for u in range(len(sf.pos_col)):
        if sf.pos_col[u] < 0:
            continue
        p = res.x[pos_of[int(sf.pos_col[u])]] if int(sf.pos_col[u]) in pos_of else 0.0
        q = res.x[pos_of[int(sf.neg_col[u])]] if int(sf.neg_col[u]) in pos_of else 0.0
        out[u] = p - q
    return out


def solve_model(model: dict, max_iter: int = 1_000_000) -> SolveResult:
    sf = build_standard_form(model)
    res = simplex_solve(sf, max_iter=max_iter)
    res.values = solution_values(sf, res)
    return res

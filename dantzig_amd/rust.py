"""Drop-in for the reference's compiled extension module `dantzig.rust`.

The reference exposes five PyO3 classes and one function (src/lib.rs:29-38,
src/pyobjs.rs:10-175).  This module offers the same names, constructor signatures, getters
and operator behaviour, backed by the HIP engine through the C ABI
(include/dantzig_amd.h, dzg_model_solve) instead of the Rust simplex:

    Variable(*, lb, ub)            .id .lb .ub                      src/pyobjs.rs:10-38
    PyLinExpr(coefs, vars)         map_ids_to_coefs, -e, e+e, e*k   src/pyobjs.rs:40-112
    PyAffExpr(*, linexpr, constant) .pylinexpr .constant            src/pyobjs.rs:114-133
    PyInequality(*, linexpr, b)                                     src/pyobjs.rs:135-152
    PySolution                     .objective_value, [Variable]     src/pyobjs.rs:154-175
    solve(objective, constraints)  -> PySolution                    src/lib.rs:16-27
"""
from __future__ import annotations

import ctypes as C
import itertools
import threading
import warnings

import numpy as np

from . import _ffi
from .exceptions import InfeasibleError, NearTieWarning, UnboundedError

_counter = itertools.count()          # static COUNTER: AtomicUsize, src/pyobjs.rs:8
_counter_lock = threading.Lock()

# options applied by solve(); see set_options()
_options: dict = {}


def set_options(**opts) -> None:
    """Engine options for subsequent solve() calls (fields of dzg_opts), e.g.
    set_options(numerics=_ffi.STRICT).  The reference has no such knob."""
    _ffi.default_opts(**opts)  # validates names
    _options.clear()
    _options.update(opts)


def _opt_float(v, what: str):
    if v is None:
        return None
    if not isinstance(v, (int, float)):
        raise TypeError(f"{what} must be a float or None")
    return float(v)


class Variable:
    __slots__ = ("_id", "_lb", "_ub")

    def __init__(self, *, lb, ub):
        self._lb = _opt_float(lb, "lb")
        self._ub = _opt_float(ub, "ub")
        with _counter_lock:
            self._id = next(_counter)

    id = property(lambda self: self._id)
    lb = property(lambda self: self._lb)
    ub = property(lambda self: self._ub)

    def __repr__(self) -> str:
        return f"rust.Variable(id={self._id}, lb={self._lb}, ub={self._ub})"


def _check_number(k, what: str) -> float:
    if not isinstance(k, (int, float)):
        raise TypeError(f"{what} must be an int or a float")
    return float(k)


class PyLinExpr:
    """sum coef_i * var_i, terms kept in first-seen order (the order defines the LP's
    column order and therefore its tie-breaks, src/simplex.rs:168-176)."""
    __slots__ = ("coefs", "vars", "_slot")

    def __init__(self, coefs, vars):
        self.coefs = [_check_number(c, "coefficient") for c in coefs]
        self.vars = list(vars)
        if len(self.coefs) != len(self.vars):
            raise ValueError("coefs and vars differ in length")
        for v in self.vars:
            if not isinstance(v, Variable):
                raise TypeError("vars must hold rust.Variable objects")
        self._slot = {v.id: i for i, v in enumerate(self.vars)}

    def map_ids_to_coefs(self) -> dict:
        return {v.id: c for c, v in zip(self.coefs, self.vars)}

    def __neg__(self) -> "PyLinExpr":
        return PyLinExpr([-c for c in self.coefs], self.vars)

    def __add__(self, other: "PyLinExpr") -> "PyLinExpr":
        if not isinstance(other, PyLinExpr):
            return NotImplemented
        coefs, vars_ = list(self.coefs), list(self.vars)
        slot = dict(self._slot)
        for c, v in zip(other.coefs, other.vars):   # merge by id, src/pyobjs.rs:86-98
            i = slot.get(v.id)
            if i is None:
                slot[v.id] = len(vars_)
                vars_.append(v)
                coefs.append(c)
            else:
                coefs[i] += c
        out = PyLinExpr.__new__(PyLinExpr)
        out.coefs, out.vars, out._slot = coefs, vars_, slot
        return out

    def __mul__(self, constant) -> "PyLinExpr":
        k = _check_number(constant, "multiplier")
        return PyLinExpr([k * c for c in self.coefs], self.vars)


class PyAffExpr:
    __slots__ = ("_linexpr", "_constant")

    def __init__(self, *, linexpr: PyLinExpr, constant):
        if not isinstance(linexpr, PyLinExpr):
            raise TypeError("linexpr must be a PyLinExpr")
        self._linexpr = linexpr
        self._constant = _check_number(constant, "constant")

    pylinexpr = property(lambda self: self._linexpr)
    constant = property(lambda self: self._constant)


class PyInequality:
    """linexpr <= b"""
    __slots__ = ("_linexpr", "_b")

    def __init__(self, *, linexpr: PyLinExpr, b):
        if not isinstance(linexpr, PyLinExpr):
            raise TypeError("linexpr must be a PyLinExpr")
        self._linexpr = linexpr
        self._b = _check_number(b, "b")


class PySolution:
    __slots__ = ("_objective_value", "_values", "iterations", "numerics", "shape")

    def __init__(self, objective_value: float, values: dict, iterations: int = 0,
                 numerics: str = "", shape=(0, 0)):
        self._objective_value = objective_value
        self._values = values
        self.iterations = iterations    # extras the reference does not expose
        self.numerics = numerics
        self.shape = shape

    objective_value = property(lambda self: self._objective_value)

    def __getitem__(self, variable: Variable) -> float:
        return self._values.get(variable.id, 0.0)   # src/pyobjs.rs:163-165


def lower(objective: PyAffExpr, constraints):
    """Flattens the call arguments into the arrays of dzg_model.  Returns (arrays, table)."""
    table: dict = {}
    order = []

    def slot(v: Variable) -> int:
        i = table.get(v.id)
        if i is None:
            i = table[v.id] = len(order)
            order.append(v)
        return i

    le = objective.pylinexpr
    obj_var = [slot(v) for v in le.vars]
    obj_coef = list(le.coefs)
    con_ptr, con_var, con_coef, con_b = [0], [], [], []
    for ineq in constraints:
        if not isinstance(ineq, PyInequality):
            raise TypeError("constraints must hold PyInequality objects")
        for c, v in zip(ineq._linexpr.coefs, ineq._linexpr.vars):
            con_var.append(slot(v))
            con_coef.append(c)
        con_ptr.append(len(con_var))
        con_b.append(ineq._b)
    arrays = dict(
        has_lb=np.array([v.lb is not None for v in order] + [False], dtype=np.int32),
        has_ub=np.array([v.ub is not None for v in order] + [False], dtype=np.int32),
        lb=np.array([0.0 if v.lb is None else v.lb for v in order] + [0.0]),
        ub=np.array([0.0 if v.ub is None else v.ub for v in order] + [0.0]),
        obj_var=np.array(obj_var + [0], dtype=np.int64), obj_coef=np.array(obj_coef + [0.0]),
        con_ptr=np.array(con_ptr, dtype=np.int64), con_var=np.array(con_var + [0], dtype=np.int64),
        con_coef=np.array(con_coef + [0.0]), con_b=np.array(con_b + [0.0]),
        nvars=len(order), obj_nterms=len(obj_var), ncons=len(con_b),
        obj_const=objective.constant)
    return arrays, order


def _c_model(a: dict) -> _ffi.Model:
    p = _ffi.ptr
    return _ffi.Model(a["nvars"], p(a["has_lb"]), p(a["has_ub"]), p(a["lb"]), p(a["ub"]),
                      a["obj_nterms"], p(a["obj_var"]), p(a["obj_coef"]), a["obj_const"],
                      a["ncons"], p(a["con_ptr"]), p(a["con_var"]), p(a["con_coef"]), p(a["con_b"]))


def solve(objective: PyAffExpr, constraints) -> PySolution:
    """Maximise `objective` subject to `constraints` on the GPU (src/lib.rs:16-27)."""
    if not isinstance(objective, PyAffExpr):
        raise TypeError("objective must be a PyAffExpr")
    arrays, order = lower(objective, list(constraints))
    _ffi.require_gpu()
    values = np.zeros(max(len(order), 1))
    res = _ffi.ModelResult()
    res.values = _ffi.ptr(values)
    opts = _ffi.default_opts(**_options)
    md = _c_model(arrays)
    rc = _ffi.lib().dzg_model_solve(C.byref(md), C.byref(opts), C.byref(res))
    _ffi.check(rc, "dzg_model_solve")
    if rc == _ffi.UNBOUNDED:
        raise UnboundedError("The objective is unbounded")      # src/lib.rs:24
    if rc == _ffi.INFEASIBLE:
        raise InfeasibleError("The model is infeasible")        # src/lib.rs:25
    if rc != _ffi.OPTIMAL:
        # PANIC / ITER_LIMIT / SINGULAR: the reference would panic (PanicException) or recurse
        raise RuntimeError(f"simplex terminated with status {_ffi.status_str(rc)!r} after "
                           f"{res.iterations} iterations")
    if res.near_ties > 0 and res.numerics_used != _ffi.STRICT:
        warnings.warn(
            f"{res.near_ties} of {res.iterations} pivots (the first: pivot {res.first_near_tie}) were "
            "decided within rounding distance of a tie and the model is too large to be re-solved "
            "in the reference's own arithmetic: the optimum is valid, but the vertex may differ "
            "from the one the reference implementation reports when the optimum is not unique",
            NearTieWarning, stacklevel=3)
    return PySolution(float(res.objective), {v.id: float(values[i]) for i, v in enumerate(order)},
                      int(res.iterations), "strict" if res.numerics_used == _ffi.STRICT else "fast",
                      (int(res.m), int(res.n)))

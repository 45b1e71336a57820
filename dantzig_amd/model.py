"""Modelling layer: Variable, LinExpr, AffExpr, Constraint.

The public surface and the lowering rules are those of the reference's modelling module
(python-source/dantzig/model.py, cited per rule below); the implementation is this project's
own: all arithmetic and all comparisons of the three expression kinds live in one mixin that
works on a normal form "(linear part, constant or None)".

Rules kept from the reference
  * a result is a LinExpr while no constant has been involved, an AffExpr afterwards
    (Variable + Variable -> LinExpr, Variable + 1 -> AffExpr; model.py:113-150,199-224);
  * a purely linear operand counts as constant 0.0 when it meets an affine one, so the float
    operations are the reference's (`0.0 + 3`, `0.0 - 3`, ...; model.py:186-187,279-313);
  * every comparison is taken on `lhs - rhs`:  linear part (op) -constant  (model.py:323-347);
  * `==` lowers to two opposite inequalities, `>=` to one negated `<=` (model.py:350-375);
  * Constraint objects are truthy, so a chained  a <= x <= b  keeps only  x <= b  -- Python
    evaluates `(a <= x) and (x <= b)` -- exactly like the reference (tests/test_optimize.py:67).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

from . import rust as rs

Number = Union[int, float]
Operand = Union[Number, "Variable", "LinExpr", "AffExpr"]


def _is_number(value) -> bool:
    return isinstance(value, (int, float))


def _build(linear: rs.PyLinExpr, constant: Optional[float]):
    """Normal form -> LinExpr (no constant involved so far) or AffExpr."""
    expr = LinExpr(linexpr=linear)
    return expr if constant is None else AffExpr(linexpr=expr, constant=constant)


class _Algebra:
    """+, -, *, unary -, ==, <=, >= for Variable, LinExpr and AffExpr."""

    def _normal_form(self) -> Tuple[rs.PyLinExpr, Optional[float]]:
        raise NotImplementedError

    # ---------------------------------------------------------------- conversions
    def to_linexpr(self) -> "LinExpr":
        linear, constant = self._normal_form()
        if constant is not None:
            raise TypeError("an affine expression has no purely linear form")
        return LinExpr(linexpr=linear)

    def to_affexpr(self) -> "AffExpr":
        linear, constant = self._normal_form()
        return AffExpr(linexpr=LinExpr(linexpr=linear), constant=0.0 if constant is None else constant)

    # ---------------------------------------------------------------- arithmetic
    def _combine(self, other: Operand, sign: int, what: str):
        linear, constant = self._normal_form()
        if _is_number(other):
            base = 0.0 if constant is None else constant
            return _build(linear, base + other if sign > 0 else base - other)
        if not isinstance(other, _Algebra):
            raise TypeError(f"{type(self).__name__}.{what}() does not support {type(other)}")
        other_linear, other_constant = other._normal_form()
        merged = linear + (other_linear if sign > 0 else -other_linear)
        if constant is None and other_constant is None:
            return _build(merged, None)
        mine = 0.0 if constant is None else constant
        theirs = 0.0 if other_constant is None else other_constant
        return _build(merged, mine + theirs if sign > 0 else mine - theirs)

    def __add__(self, rhs: Operand):
        return self._combine(rhs, +1, "__add__")

    def __sub__(self, rhs: Operand):
        return self._combine(rhs, -1, "__sub__")

    def __radd__(self, lhs: Number) -> "AffExpr":
        return self + lhs

    def __rsub__(self, lhs: Number) -> "AffExpr":
        return (-self) + lhs

    def __neg__(self):
        linear, constant = self._normal_form()
        return _build(-linear, None if constant is None else -constant)

    def __mul__(self, rhs: Number):
        if not _is_number(rhs):
            raise TypeError(f"{type(self).__name__}.__mul__() only supports int and float")
        linear, constant = self._normal_form()
        return _build(linear * rhs, None if constant is None else rhs * constant)

    def __rmul__(self, lhs: Number):
        return self * lhs

    # ---------------------------------------------------------------- comparisons -> constraints
    def _difference(self, rhs: Operand) -> Tuple["LinExpr", float]:
        diff = self.to_affexpr() - rhs
        return diff.linexpr, -diff.constant

    def __eq__(self, rhs: Operand) -> "Constraint":  # type: ignore[override]
        linexpr, b = self._difference(rhs)
        return Constraint.equality(linexpr=linexpr, b=b)

    def __le__(self, rhs: Operand) -> "Constraint":
        linexpr, b = self._difference(rhs)
        return Constraint.less_than_eq(linexpr=linexpr, b=b)

    def __ge__(self, rhs: Operand) -> "Constraint":
        linexpr, b = self._difference(rhs)
        return Constraint.greater_than_eq(linexpr=linexpr, b=b)


class Variable(_Algebra):
    """A decision variable with optional inclusive bounds (model.py:8-46).

    `lb` and `ub` are both required keywords; None means unbounded on that side."""

    def __init__(self, *, lb, ub, name=None) -> None:
        self._name = name
        self._variable = rs.Variable(lb=lb, ub=ub)

    @classmethod
    def free(cls, name=None) -> "Variable":
        return cls(lb=None, ub=None, name=name)

    @classmethod
    def nonneg(cls, name=None) -> "Variable":
        return cls(lb=0.0, ub=None, name=name)

    @classmethod
    def nonpos(cls, name=None) -> "Variable":
        return cls(lb=None, ub=0.0, name=name)

    nn = nonneg
    np = nonpos

    name = property(lambda self: self._name)
    id = property(lambda self: self._variable.id)
    lb = property(lambda self: self._variable.lb)
    ub = property(lambda self: self._variable.ub)

    def to_rust_variable(self) -> rs.Variable:
        return self._variable

    def _normal_form(self):
        return rs.PyLinExpr(coefs=[1.0], vars=[self._variable]), None

    def __hash__(self) -> int:
        return hash(self.id)

    def __repr__(self) -> str:
        return f"Variable(id={self.id}, lb={self.lb}, ub={self.ub})"


class LinExpr(_Algebra):
    """A linear combination of variables (model.py:171-243)."""

    __hash__ = None  # type: ignore[assignment]

    def __init__(self, *, linexpr: rs.PyLinExpr) -> None:
        self._linexpr = linexpr

    @classmethod
    def from_rust_variable(cls, variable: rs.Variable) -> "LinExpr":
        return cls(linexpr=rs.PyLinExpr(coefs=[1.0], vars=[variable]))

    def to_rust_linexpr(self) -> rs.PyLinExpr:
        return self._linexpr

    def to_linexpr(self) -> "LinExpr":
        return self

    def map_ids_to_coefs(self) -> dict:
        return self._linexpr.map_ids_to_coefs()

    def _normal_form(self):
        return self._linexpr, None


class AffExpr(_Algebra):
    """Linear part plus constant (model.py:246-347)."""

    __hash__ = None  # type: ignore[assignment]

    def __init__(self, *, linexpr: LinExpr, constant: Number) -> None:
        self._affexpr = rs.PyAffExpr(linexpr=linexpr.to_rust_linexpr(), constant=constant)

    @classmethod
    def from_rust_variable(cls, variable: rs.Variable) -> "AffExpr":
        return cls(linexpr=LinExpr.from_rust_variable(variable), constant=0.0)

    def to_rust_affexpr(self) -> rs.PyAffExpr:
        return self._affexpr

    def to_affexpr(self) -> "AffExpr":
        return self

    linexpr = property(lambda self: LinExpr(linexpr=self._affexpr.pylinexpr))
    constant = property(lambda self: self._affexpr.constant)

    def _normal_form(self):
        return self._affexpr.pylinexpr, self._affexpr.constant


class Constraint:
    """One or two rows `linexpr <= b` (model.py:350-378)."""

    def __init__(self, *, inequalities: list) -> None:
        self._inequalities = inequalities

    @staticmethod
    def _row(linexpr: LinExpr, b: Number, negate: bool) -> rs.PyInequality:
        if negate:
            return rs.PyInequality(linexpr=(-linexpr).to_rust_linexpr(), b=-b)
        return rs.PyInequality(linexpr=linexpr.to_rust_linexpr(), b=b)

    @classmethod
    def less_than_eq(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[cls._row(linexpr, b, False)])

    @classmethod
    def greater_than_eq(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[cls._row(linexpr, b, True)])

    @classmethod
    def equality(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[cls._row(linexpr, b, False), cls._row(linexpr, b, True)])

    def rust_inequalities(self) -> list:
        return self._inequalities

"""Modelling layer: Variable, LinExpr, AffExpr, Constraint.

Same public surface and lowering rules as the reference's python-source/dantzig/model.py
(cited per method), written against dantzig_amd.rust:

  * Variable + number / AffExpr -> AffExpr;  Variable + Variable / LinExpr -> LinExpr
  * every comparison is taken on  lhs - rhs  as an AffExpr:  linexpr (op) -constant
  * `==` lowers to two opposite inequalities, `>=` to one negated `<=`   (model.py:323-375)
  * Constraint objects are truthy, so a chained  a <= x <= b  keeps only  x <= b
    (Python evaluates `(a <= x) and (x <= b)`), exactly like the reference
    (tests/test_optimize.py:67).
"""
from __future__ import annotations

from typing import Union

from . import rust as rs

Number = Union[int, float]


def _is_number(v) -> bool:
    return isinstance(v, (int, float))


class _Algebra:
    """Operators shared by Variable, LinExpr and AffExpr.  Subclasses provide to_affexpr()
    and, when they are purely linear, to_linexpr()."""

    def to_affexpr(self) -> "AffExpr":
        raise NotImplementedError

    # ---- comparisons build constraints (model.py:152-165, 226-239, 323-347)
    def __eq__(self, rhs) -> "Constraint":  # type: ignore[override]
        diff = self.to_affexpr() - rhs
        return Constraint.equality(linexpr=diff.linexpr, b=-diff.constant)

    def __le__(self, rhs) -> "Constraint":
        diff = self.to_affexpr() - rhs
        return Constraint.less_than_eq(linexpr=diff.linexpr, b=-diff.constant)

    def __ge__(self, rhs) -> "Constraint":
        diff = self.to_affexpr() - rhs
        return Constraint.greater_than_eq(linexpr=diff.linexpr, b=-diff.constant)

    def __radd__(self, lhs: Number) -> "AffExpr":
        return self + lhs

    def __rmul__(self, lhs: Number):
        return self * lhs


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

    def to_linexpr(self) -> "LinExpr":
        return LinExpr.from_rust_variable(self._variable)

    def to_affexpr(self) -> "AffExpr":
        return AffExpr.from_rust_variable(self._variable)

    def __add__(self, rhs):
        return self.to_linexpr() + rhs

    def __sub__(self, rhs):
        return self.to_linexpr() - rhs

    def __rsub__(self, lhs: Number) -> "AffExpr":
        return -self.to_linexpr() + lhs

    def __mul__(self, rhs: Number) -> "LinExpr":
        if not _is_number(rhs):
            raise TypeError("Variable.__mul__() only supports int and float")
        return self.to_linexpr() * rhs

    def __neg__(self) -> "LinExpr":
        return -self.to_linexpr()

    def __hash__(self) -> int:
        return hash(self.id)

    def __repr__(self) -> str:
        return f"Variable(id={self.id}, lb={self.lb}, ub={self.ub})"


class LinExpr(_Algebra):
    """A linear combination of variables (model.py:171-243)."""

    def __init__(self, *, linexpr: rs.PyLinExpr) -> None:
        self._linexpr = linexpr

    @classmethod
    def from_rust_variable(cls, variable: rs.Variable) -> "LinExpr":
        return cls(linexpr=rs.PyLinExpr(coefs=[1.0], vars=[variable]))

    def to_rust_linexpr(self) -> rs.PyLinExpr:
        return self._linexpr

    def to_linexpr(self) -> "LinExpr":
        return self

    def to_affexpr(self) -> "AffExpr":
        return AffExpr(linexpr=self, constant=0.0)

    def map_ids_to_coefs(self) -> dict:
        return self._linexpr.map_ids_to_coefs()

    def __add__(self, rhs):
        if _is_number(rhs) or isinstance(rhs, AffExpr):
            return self.to_affexpr() + rhs
        if isinstance(rhs, (Variable, LinExpr)):
            return LinExpr(linexpr=self._linexpr + rhs.to_linexpr()._linexpr)
        raise TypeError(f"LinExpr.__add__() does not support {type(rhs)}")

    def __sub__(self, rhs):
        if _is_number(rhs) or isinstance(rhs, AffExpr):
            return self.to_affexpr() - rhs
        if isinstance(rhs, (Variable, LinExpr)):
            return self + (-rhs.to_linexpr())
        raise TypeError(f"LinExpr.__sub__() does not support {type(rhs)}")

    def __rsub__(self, lhs: Number) -> "AffExpr":
        return -self + lhs

    def __mul__(self, rhs: Number) -> "LinExpr":
        if not _is_number(rhs):
            raise TypeError("LinExpr.__mul__() only supports int and float")
        return LinExpr(linexpr=self._linexpr * rhs)

    def __neg__(self) -> "LinExpr":
        return LinExpr(linexpr=-self._linexpr)

    __hash__ = None  # type: ignore[assignment]


class AffExpr(_Algebra):
    """linear part + constant (model.py:246-347)."""

    def __init__(self, *, linexpr: LinExpr, constant: Number) -> None:
        self._affexpr = rs.PyAffExpr(linexpr=linexpr.to_rust_linexpr(), constant=constant)

    @classmethod
    def from_rust_variable(cls, variable: rs.Variable) -> "AffExpr":
        return cls(linexpr=LinExpr.from_rust_variable(variable), constant=0.0)

    def to_rust_affexpr(self) -> rs.PyAffExpr:
        return self._affexpr

    def to_affexpr(self) -> "AffExpr":
        return self

    @property
    def linexpr(self) -> LinExpr:
        return LinExpr(linexpr=self._affexpr.pylinexpr)

    @property
    def constant(self) -> float:
        return self._affexpr.constant

    def __add__(self, rhs) -> "AffExpr":
        if _is_number(rhs):
            return AffExpr(linexpr=self.linexpr, constant=self.constant + rhs)
        if isinstance(rhs, (Variable, LinExpr, AffExpr)):
            other = rhs.to_affexpr()
            return AffExpr(linexpr=self.linexpr + other.linexpr,
                           constant=self.constant + other.constant)
        raise TypeError(f"AffExpr.__add__() does not support {type(rhs)}")

    def __sub__(self, rhs) -> "AffExpr":
        if _is_number(rhs):
            return AffExpr(linexpr=self.linexpr, constant=self.constant - rhs)
        if isinstance(rhs, (Variable, LinExpr, AffExpr)):
            other = rhs.to_affexpr()
            return AffExpr(linexpr=self.linexpr - other.linexpr,
                           constant=self.constant - other.constant)
        raise TypeError(f"AffExpr.__sub__() does not support {type(rhs)}")

    def __rsub__(self, lhs: Number) -> "AffExpr":
        return -self + lhs

    def __mul__(self, rhs: Number) -> "AffExpr":
        if not _is_number(rhs):
            raise TypeError("AffExpr.__mul__() only supports int and float")
        return AffExpr(linexpr=rhs * self.linexpr, constant=rhs * self.constant)

    def __neg__(self) -> "AffExpr":
        return AffExpr(linexpr=-self.linexpr, constant=-self.constant)

    __hash__ = None  # type: ignore[assignment]


class Constraint:
    """One or two `linexpr <= b` rows (model.py:350-378)."""

    def __init__(self, *, inequalities: list) -> None:
        self._inequalities = inequalities

    @classmethod
    def less_than_eq(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[rs.PyInequality(linexpr=linexpr.to_rust_linexpr(), b=b)])

    @classmethod
    def greater_than_eq(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[rs.PyInequality(linexpr=(-linexpr).to_rust_linexpr(), b=-b)])

    @classmethod
    def equality(cls, *, linexpr: LinExpr, b: Number) -> "Constraint":
        return cls(inequalities=[
            rs.PyInequality(linexpr=linexpr.to_rust_linexpr(), b=b),
            rs.PyInequality(linexpr=(-linexpr).to_rust_linexpr(), b=-b),
        ])

    def rust_inequalities(self) -> list:
        return self._inequalities

"""Minimize / Maximize / Solution (python-source/dantzig/optimize.py:8-154).

The core always maximises: Minimize negates the objective on the way in and the optimal value
on the way out (optimize.py:114-117, :21-27)."""
from __future__ import annotations

import abc
from typing import Union

from . import rust as rs
from .model import AffExpr, Constraint, LinExpr, Variable

_SENSES = ("minimize", "maximize")


class Solution:
    def __init__(self, *, solution: rs.PySolution, sense: str) -> None:
        if sense not in _SENSES:
            raise ValueError(f"sense is {sense!r}; a Solution is built for {_SENSES[0]!r} or {_SENSES[1]!r}")
        self._solution = solution
        self._sense = sense

    @property
    def objective_value(self) -> float:
        value = self._solution.objective_value
        return -value if self._sense == "minimize" else value

    def __getitem__(self, variable: Variable) -> float:
        return self._solution[variable.to_rust_variable()]


class Optimize(abc.ABC):
    """Common part of Minimize and Maximize: objective, constraint list, chaining."""

    def __init__(self, objective: Union[Variable, LinExpr, AffExpr]) -> None:
        self.objective = objective.to_affexpr()
        self.constraints: list = []

    @property
    @abc.abstractmethod
    def sense(self) -> str:
        raise NotImplementedError

    def subject_to(self, constraints):
        """Add one constraint or a list of constraints; returns self for chaining."""
        batch = [constraints] if isinstance(constraints, Constraint) else constraints
        if not isinstance(batch, list):  # like the reference, list items are not inspected here
            raise TypeError("subject_to takes a Constraint or a list of Constraints, got "
                            f"{type(constraints).__name__}")
        self.constraints += batch
        return self

    st = subject_to

    def yield_rust_inequalities(self):
        for constraint in self.constraints:
            yield from constraint.rust_inequalities()

    def _core_objective(self) -> AffExpr:
        return -self.objective if self.sense == "minimize" else self.objective

    def solve(self) -> Solution:
        """Solve on the GPU.  Raises exceptions.UnboundedError / InfeasibleError."""
        objective = self._core_objective().to_rust_affexpr()
        inequalities = list(self.yield_rust_inequalities())
        return Solution(solution=rs.solve(objective, inequalities), sense=self.sense)


class Minimize(Optimize):
    """min objective  s.t. constraints.

    >>> x = Variable(lb=1.0, ub=None); y = Variable(lb=None, ub=2.0)
    >>> result = Minimize(x - 5 * y).solve()      # result[x] == 1.0, result[y] == 2.0
    """

    sense = property(lambda self: "minimize")


class Maximize(Optimize):
    """max objective  s.t. constraints."""

    sense = property(lambda self: "maximize")

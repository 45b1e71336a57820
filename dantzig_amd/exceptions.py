"""Solver outcome exceptions (same names and meaning as dantzig.exceptions)."""


class UnboundedError(Exception):
    """The objective can be improved without limit."""


class InfeasibleError(Exception):
    """No point satisfies all constraints."""

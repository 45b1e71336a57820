"""Exceptions raised by `.solve()` -- the two outcomes of the reference's `Error` enum
(src/error.rs:3-7), under the names the reference's Python package uses."""


class SolveError(Exception):
    """A solve ended without an optimal vertex."""


class UnboundedError(SolveError):
    """The objective can be improved without limit (src/simplex.rs:313)."""


class InfeasibleError(SolveError):
    """No point satisfies all constraints (src/simplex.rs:325)."""


class NearTieWarning(UserWarning):
    """FAST numerics met a pivot choice within rounding of a tie on a model too large for the
    bit-exact re-solve (not in the reference: it has one arithmetic only)."""

"""dantzig_amd: an MI355X-native parametric self-dual simplex core behind dantzig's
modelling surface.

    import dantzig_amd as dz
    x, y, z = dz.Variable.nonneg(), dz.Variable.nonneg(), dz.Variable.nonneg()
    sol = dz.Minimize(x + y - z).subject_to(x + y + z == 1).solve()

Public names match the reference package (python-source/dantzig/__init__.py:2-10).
`.solve()` runs on the GPU through the C ABI in include/dantzig_amd.h; there is no CPU path.
"""
from . import exceptions
from .model import Variable
from .optimize import Maximize, Minimize

Var = Variable
Min = Minimize
Max = Maximize

__all__ = ["Variable", "Var", "Minimize", "Min", "Maximize", "Max", "exceptions"]

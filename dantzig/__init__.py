"""`import dantzig` resolves to the MI355X-native implementation in `dantzig_amd`.

The reference ships a package of this name whose arithmetic lives in the compiled extension
`dantzig.rust` (Cargo.toml:12-14, src/lib.rs:29-38; imported at
python-source/dantzig/model.py:5 and optimize.py:4).  This alias lets code written against
the reference -- `import dantzig as dz`, `from dantzig import rust`, `dantzig.exceptions` --
run on the HIP engine unchanged.  It contains no logic of its own: every name is the object
defined in `dantzig_amd`.
"""
import sys as _sys

import dantzig_amd as _impl
from dantzig_amd import (Max, Maximize, Min, Minimize, Var, Variable, exceptions, model,  # noqa: F401
                         optimize, rust)

# submodules under the reference's names: `import dantzig.rust`, `from dantzig.model import ...`
for _name in ("rust", "model", "optimize", "exceptions"):
    _sys.modules[__name__ + "." + _name] = getattr(_impl, _name)

__all__ = list(_impl.__all__)

"""Find Level-2 models (> auto_strict_rows rows) on which FAST alone gives up, and check that AUTO
then answers like STRICT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dantzig_amd as dz
from dantzig_amd import rust, _ffi

def model(seed, ncons=130, nvars=110):
    rng = np.random.default_rng(seed)
    a = (rng.uniform(size=(ncons, nvars)) < 0.25).astype(float)
    b = rng.integers(0, 4, ncons).astype(float)
    c = rng.integers(-1, 6, nvars).astype(float)
    xs = [dz.Variable.nonneg() for _ in range(nvars)]
    obj = sum((float(c[j]) * xs[j] for j in range(1, nvars)), float(c[0]) * xs[0])
    cons = []
    for i in range(ncons):
        nz = np.nonzero(a[i])[0]
        if len(nz) == 0:
            continue
        cons.append(sum((xs[j] * 1.0 for j in nz[1:]), xs[nz[0]] * 1.0) <= float(b[i]))
    return dz.Maximize(obj).subject_to(cons)

def run(seed, numerics):
    rust.set_options(numerics=numerics)
    try:
        sol = model(seed).solve()
        return ("optimal", sol.objective_value)
    except Exception as exc:  # noqa: BLE001
        return (type(exc).__name__, str(exc)[:60])
    finally:
        rust.set_options()

for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    fast, strict, auto = run(seed, _ffi.FAST), run(seed, _ffi.STRICT), run(seed, _ffi.AUTO)
    print(seed, "FAST", fast, "| STRICT", strict, "| AUTO", auto, "| auto==strict", auto == strict, flush=True)

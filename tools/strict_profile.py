"""A few STRICT pivots at one size, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 6
a, b, c = core.gen_dense_lp(seed=1002, m=m, n_struct=2 * m)
lp = core.CoreLP.from_inequality_form(a, b, c)
with core.Solver(lp, numerics=core.STRICT) as s:
    s.run(2)
    t = time.perf_counter(); s.run(pivots); dt = time.perf_counter() - t
print(f"STRICT {m}: {1e3 * dt / pivots:.2f} ms/pivot")

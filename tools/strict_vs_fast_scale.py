"""One-off parity evidence at the benchmark size: first pivots of the 8192x16384 LP in STRICT
(reference arithmetic on the GPU, bit-identical to the oracle at every size both can run) and
in FAST numerics.  Output is committed under profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core
m, ns, seed, pivots = 8192, 16384, 1003, int(sys.argv[1]) if len(sys.argv) > 1 else 16
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
t = time.time(); strict = core.solve(lp, numerics=core.STRICT, max_iter=pivots); ts = time.time() - t
t = time.time(); fast = core.solve(lp, numerics=core.FAST, max_iter=pivots); tf = time.time() - t
ls = [(k, e, l) for k, e, l, _ in strict.pivots]; lf = [(k, e, l) for k, e, l, _ in fast.pivots]
print(f"{m}x{ns} seed {seed}: {pivots} pivots  STRICT {ts:.1f}s  FAST {tf:.2f}s")
print("pivot logs identical:", ls == lf)
print("max |mu_fast - mu_strict| / |mu|:", max(abs(p[3] - q[3]) / abs(q[3]) for p, q in zip(fast.pivots, strict.pivots)))
print("max |x_fast - x_strict|:", float(np.abs(fast.x - strict.x).max()))
print("log:", ls)

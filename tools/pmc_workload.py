"""Short workload for PMC passes: 2500 FAST pivots on the 8192x16384 LP (39 MFMA flushes) and one
on-device refactorisation (blocked LU + MFMA GEMMs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
a, b, c = core.gen_dense_lp(seed=1003, m=8192, n_struct=16384)
lp = core.CoreLP.from_inequality_form(a, b, c)
with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=50) as s:
    s.run(2500)
    s.refactor()
    s.run(100)
    r = s.result(log=False)
    print(r.status, r.iterations, r.max_pivot_error)

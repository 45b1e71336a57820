"""All P ranks of a column-sharded FAST solve in one process on one GPU (device-copy exchange),
for per-rank kernel timing under rocprofv3:  per-rank compute per pivot = sum of kernel time / P.

  python3 tools/lockstep_profile.py [P] [pivots] [rows] [cols] [seed] [replicate 0|1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core, sharded  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1003
replicate = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True

a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
lp = core.CoreLP.from_inequality_form(a, b, c)
solvers = sharded.make_lockstep(lp, P, replicate=replicate, poll_interval=50)
sharded.run_lockstep(solvers, 100)
solvers[0].poll()
t0 = time.perf_counter()
st = sharded.run_lockstep(solvers, pivots)
solvers[0].poll()
dt = time.perf_counter() - t0
print(f"P={P} {rows}x{cols} ({'replicated matrix, 64-byte records' if replicate else 'partitioned matrix'}): {pivots} pivots, status {st}, {1e6 * dt / pivots:.1f} us per pivot for all "
      f"{P} ranks on one GPU = {1e6 * dt / pivots / P:.1f} us per rank per pivot (no communication)")
for s in solvers:
    s.close()

"""All P ranks of a column-sharded FAST solve in one process on one GPU (device-copy exchange),
for per-rank kernel timing:  per-rank compute per pivot = wall time of the lockstep loop / P (the
ranks' kernels run one behind the other on one stream; no communication cost).

  python3 tools/lockstep_profile.py [P] [pivots] [rows] [cols] [seed] [replicate 0|1] [warm_k] [shard_rows 0|1]

warm_k > 0: the solve starts from a basis of warm_k structural columns (core.warm_started), the
deep regime of a solve; P = 1: the plain single-GPU solver on the same state (the denominator).
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
warm_k = int(sys.argv[7]) if len(sys.argv) > 7 else 0
shard_rows = bool(int(sys.argv[8])) if len(sys.argv) > 8 else False

a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
lp = core.CoreLP.from_inequality_form(a, b, c)
if warm_k:
    lp = core.warm_started(lp, warm_k)
what = f"{rows}x{cols}" + (f" warm-started at k = {warm_k}" if warm_k else "")
if P == 1:
    with core.Solver(lp, numerics=core.FAST, poll_interval=50) as s:
        s.run(100)
        t0 = time.perf_counter()
        st = s.run(pivots)
        dt = time.perf_counter() - t0
        k = s.result(log=False).dense_columns
    print(f"P=1 {what} (one GPU, the whole solver): {pivots} pivots, status {st}, k = {k}, "
          f"{1e6 * dt / pivots:.1f} us per pivot")
    sys.exit(0)
solvers = sharded.make_lockstep(lp, P, replicate=replicate, shard_rows=shard_rows, poll_interval=50)
sharded.run_lockstep(solvers, 100)
solvers[0].poll()
t0 = time.perf_counter()
st = sharded.run_lockstep(solvers, pivots)
solvers[0].poll()
dt = time.perf_counter() - t0
k = solvers[0].result(log=False).dense_columns
mode = ("replicated matrix" if replicate else "partitioned matrix") + (
    ", basis side sharded by rows" if shard_rows else ", basis side replicated")
print(f"P={P} {what} ({mode}): {pivots} pivots, status {st}, k = {k}, {1e6 * dt / pivots:.1f} us per pivot for all "
      f"{P} ranks on one GPU = {1e6 * dt / pivots / P:.1f} us per rank per pivot (no communication)")
for s in solvers:
    s.close()

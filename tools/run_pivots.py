"""Run the first N pivots of a G1 LP with the FAST engine and stop (for profiling a regime of the
solve: two rocprofv3 --stats runs with N and N + d pivots differ by the d pivots at depth N).

  python3 tools/run_pivots.py <pivots> [rows] [cols] [seed] [sparse_per_col] [warm_k]

warm_k > 0 (dense): start from a basis of warm_k structural columns (core.warm_started) instead of the
slack basis -- a regime of the solve (compact width k) without the pivots that lead there.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core  # noqa: E402

pivots = int(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1003
per_col = int(sys.argv[5]) if len(sys.argv) > 5 else 0
warm_k = int(sys.argv[6]) if len(sys.argv) > 6 else 0
if per_col > 0:
    cp, ri, val, b, c = core.gen_sparse_lp(seed, rows, cols, per_col)
    lp = core.CoreLP.from_csc(rows, cp, ri, val, b, c)
else:
    a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    if warm_k > 0:
        lp = core.warm_started(lp, warm_k)
with core.Solver(lp, numerics=core.FAST, poll_interval=50, log_capacity=1) as s:
    t0 = time.perf_counter()
    status = s.run(pivots)
    dt = time.perf_counter() - t0
    r = s.result(log=False)
print(f"{rows}x{cols} seed {seed}: {status} after {r.iterations} pivots in {dt:.2f} s, k = {r.dense_columns}, "
      f"max_pivot_error {r.max_pivot_error:.2e}")

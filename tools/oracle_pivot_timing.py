"""Time real pivots of the CPU oracle (the C restatement of the reference: a dense LU of B and
of B^T from scratch in every iteration) at benchmark size, so that bench.py's modelled
`cpu_baseline` can cite a measurement beside it ("config 3: first 2
pivots timed").  Needs no GPU: the LP comes from the library's host-side generator G1.

  python3 tools/oracle_pivot_timing.py [rows (8192)] [cols (16384)] [seed (1003)] [pivots (2)]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as ora
from dantzig_amd import core

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1003
pivots = int(sys.argv[4]) if len(sys.argv) > 4 else 2

t0 = time.perf_counter()
a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
sf = ora.stdform_from_dense(a, b, c)
print(f"G1 {rows}x{cols} seed {seed}: generated and put in standard form in "
      f"{time.perf_counter() - t0:.1f} s", flush=True)
prev = 0.0
for k in range(1, pivots + 1):
    t0 = time.perf_counter()
    res = ora.simplex_solve(sf, max_iter=k, log_cap=k)
    dt = time.perf_counter() - t0
    print(f"first {res.iterations} pivot(s): {dt:.1f} s of CPU on 1 core "
          f"(+{dt - prev:.1f} s for pivot {k}); log {res.pivots}", flush=True)
    prev = dt
print(f"=> {(prev / pivots):.1f} s per pivot, {pivots / prev:.5f} iterations/s at {rows} rows "
      f"(host: {os.cpu_count()} cores, 1 used)")

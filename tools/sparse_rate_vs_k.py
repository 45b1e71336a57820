"""BASELINE config 4 (sparse 50 000 x 100 000, 50 nonzeros per column, CSC on the device) deep into
its solve: pivots per second and the size k of the basis block every `chunk` pivots, until k
passes `k_stop` or `max_seconds` elapse.  Shows whether the rate survives the growth of the basis
(the dense-inverse representation of round 1 paid 8*m*k bytes per FTRAN).

  python3 tools/sparse_rate_vs_k.py [k_stop (12000)] [chunk (5000)] [max_seconds (240)] [rows cols per_col seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import _ffi, core

k_stop = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
max_s = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
m, ns, per_col, seed = (int(a) for a in sys.argv[4:8]) if len(sys.argv) > 7 else (50000, 100000, 50, 1004)
cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
t_all = time.time()
with core.Solver(lp, numerics=core.FAST, poll_interval=50, profile=1 << _ffi.K_PRICE) as s:
    print(f"sparse {m}x{ns}, {per_col} per column, seed {seed}: pivots, k, it/s over the last "
          f"{chunk}, pricing us/launch, max_pivot_error, near ties, refactors", flush=True)
    prev = s.result(log=False)
    status = "iter_limit"
    while status == "iter_limit" and prev.dense_columns < k_stop and time.time() - t_all < max_s:
        t0 = time.time()
        status = s.run(chunk)
        dt = time.time() - t0
        r = s.result(log=False)
        n = r.iterations - prev.iterations
        us = 1e3 * (r.kernel_ms["price"] - prev.kernel_ms["price"]) / max(
            r.kernel_launches["price"] - prev.kernel_launches["price"], 1)
        print(f"  {r.iterations:8d}  k={r.dense_columns:6d}  {n / dt:8.0f} it/s  {us:6.1f} us  "
              f"{r.max_pivot_error:.1e}  {r.near_ties}  {r.refactors}", flush=True)
        prev = r
    print("status:", status, f" total {time.time() - t_all:.0f} s")

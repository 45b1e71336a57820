"""Replay a complete oracle pivot log (tests/golden/oracle_pivots_*.npz) with STRICT numerics on
the GPU and compare BIT FOR BIT: kind, entering, leaving and mu of every pivot, the final basis
and the objective.  STRICT costs 11 ms (512 rows) to 33 ms (1024 rows) per pivot, so this is a
one-off whose output is committed under profiles/; the GPU suite replays the same logs with FAST.

  python3 tools/strict_replay_oracle_log.py tests/golden/oracle_pivots_2001_512x1024.npz
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core

fx = np.load(sys.argv[1])
seed, m, ns = int(fx["seed"]), int(fx["m"]), int(fx["n_struct"])
total = int(fx["iterations"])
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
t = time.time()
with core.Solver(lp, numerics=core.STRICT) as s:
    status = "iter_limit"
    while status == "iter_limit":
        status = s.run(500)
        done = s.result(log=False).iterations
        print(f"  STRICT {done}/{total} pivots, {time.time() - t:.0f} s", flush=True)
    got = s.result()
dt = time.time() - t
kinds = np.array([p[0] for p in got.pivots]); enter = np.array([p[1] for p in got.pivots])
leave = np.array([p[2] for p in got.pivots]); mu = np.array([p[3] for p in got.pivots])
print(f"{m}x{ns} seed {seed}: STRICT {got.status} after {got.iterations} pivots in {dt:.0f} s "
      f"({1e3 * dt / max(got.iterations, 1):.1f} ms/pivot); oracle: {fx['status']} after {total} "
      f"({float(fx['oracle_seconds']):.0f} s of CPU)")
print("kind / entering / leaving identical:",
      bool(np.array_equal(kinds, fx["kind"]) and np.array_equal(enter, fx["entering"])
           and np.array_equal(leave, fx["leaving"])))
print("mu bit-identical on every pivot:", bool(np.array_equal(mu.view(np.int64), fx["mu"].view(np.int64))))
print("final basis identical:", bool(np.array_equal(got.basis, fx["basis"])))
print("objective bit-identical:", got.objective == float(fx["objective"]), repr(got.objective))

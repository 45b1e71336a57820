"""FAST against STRICT (the reference's arithmetic on the GPU) on the LP of
profiles/r02_near_tie_census_continuous_lps.txt that meets a near tie -- 2048 x 4096, G1 seed 2002 --
with the carried-state drift measured at every refactorisation (csrc/k_drift.hip) and part of the
tie tolerance: tau = max(tie_tol, 64 max_pivot_error, 4 state_drift).

FAST runs first (seconds), in chunks of `interval` pivots with a refactorisation at every chunk
boundary, so that state_drift is read after each one; STRICT follows in chunks of 1 000 pivots
(57 ms per pivot) and every chunk is compared with FAST's log as soon as it is there: an
interrupted run keeps what it has.

  python3 tools/drift_vs_strict.py [pivots] [rows] [cols] [seed] [interval]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from dantzig_amd import core  # noqa: E402

pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 2002
interval = int(sys.argv[5]) if len(sys.argv) > 5 else 5000

a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
print(f"{m}x{ns} seed {seed}: {pivots} pivots, FAST refactorises every {interval}", flush=True)
t = time.time()
with core.Solver(lp, numerics=core.FAST, poll_interval=50, refactor_interval=interval, max_iter=pivots) as s:
    done = 0
    while done < pivots:
        s.run(min(interval, pivots - done) + (50 if done + interval < pivots else 0))  # (past the boundary: the refactorisation runs)
        r = s.result(log=False)
        done = r.iterations
        print(f"  FAST {done} pivots: refactors {r.refactors}, state_drift {r.state_drift:.3e}, max_pivot_error "
              f"{r.max_pivot_error:.2e}, near_ties {r.near_ties} (first {r.first_near_tie}), min_margin {r.min_margin:.3e}",
              flush=True)
        if r.status != "iter_limit":
            break
    fast = s.result()
print(f"FAST: {fast.status} after {fast.iterations} pivots in {time.time() - t:.1f} s", flush=True)
lf = [(k, e, l) for k, e, l, _ in fast.pivots]
t = time.time()
agree = True
with core.Solver(lp, numerics=core.STRICT, max_iter=pivots) as s:
    done, status = 0, "iter_limit"
    while status == "iter_limit" and done < pivots:
        status = s.run(min(1000, pivots - done))
        r = s.result()
        ls = [(k, e, l) for k, e, l, _ in r.pivots]
        n = min(len(ls), len(lf))
        same = ls[:n] == lf[:n]
        gap = max(abs(p[3] - q[3]) / max(1.0, abs(q[3])) for p, q in zip(fast.pivots[done:n], r.pivots[done:n])) if n > done else 0.0
        first = next((i for i, (p, q) in enumerate(zip(ls[:n], lf[:n])) if p != q), -1)
        print(f"  STRICT {len(ls)} pivots, {time.time() - t:.0f} s: logs identical so far {same}"
              f"{'' if same else f' (first difference at pivot {first})'}; mu gap of this chunk {gap:.2e}", flush=True)
        agree = agree and same
        done = len(ls)
        if not same:
            break
print(f"STRICT {done} pivots in {time.time() - t:.0f} s; FAST = STRICT over them: {agree}; "
      f"FAST flagged {fast.near_ties} decisions (first at pivot {fast.first_near_tie}), final state_drift {fast.state_drift:.3e}")

"""How often does FAST numerics meet a decision inside the tie tolerance on continuous data?
Whole FAST solves (near ties counted, not stopped) of G1 LPs; AUTO would hand an LP of up to 2048
rows to STRICT at the first such pivot, so this is the rate of those hand-overs.

  python3 tools/near_tie_census.py [rows:cols:seed ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core

cases = sys.argv[1:] or ["512:1024:2001", "1024:2048:1002", "1024:2048:11", "1024:2048:12", "1024:2048:13",
                         "2048:4096:2002", "2048:4096:21", "1500:1000:31", "700:3000:32"]
for spec in cases:
    m, ns, seed = (int(t) for t in spec.split(":"))
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    t0 = time.time()
    with core.Solver(lp, numerics=core.FAST, poll_interval=256) as sv:
        sv.run(0)
        r = sv.result(log=True, log_cap=1 << 22)
    flagged = np.flatnonzero(~(r.margins > 1e-11)) if r.margins is not None else []
    print(f"{m}x{ns} seed {seed}: {r.status} after {r.iterations} pivots in {time.time() - t0:.1f} s; near ties "
          f"{r.near_ties} (first at pivot {r.first_near_tie}), min margin {r.min_margin:.2e}, "
          f"max_pivot_error {r.max_pivot_error:.1e}; margins <= 1e-11 at pivots {list(flagged[:8])}", flush=True)

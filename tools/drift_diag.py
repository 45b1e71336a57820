"""Diagnostic: max_pivot_error, refactors and near ties of a FAST solve every `chunk` pivots.
  python3 tools/drift_diag.py [rows] [cols] [seed] [chunk] [chunks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
m, ns, seed, chunk, chunks = (int(a) for a in (sys.argv[1:6] + ["4096", "8192", "1006", "250", "40"][len(sys.argv) - 1:]))
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
with core.Solver(lp, numerics=core.FAST, poll_interval=50) as s:
    for i in range(chunks):
        st = s.run(chunk)
        r = s.result(log=True, log_cap=1 << 20)
        slack_in = sum(1 for p in r.pivots[-chunk:] if p[1] >= ns)
        print(f"{r.iterations:7d} {st:10s} k={r.dense_columns:5d} err={r.max_pivot_error:.2e} refactors={r.refactors} "
              f"near_ties={r.near_ties} slacks entering in this chunk: {slack_in}", flush=True)
        if st != "iter_limit":
            break

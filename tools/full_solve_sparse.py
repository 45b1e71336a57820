"""Whole FAST solve of a G2 sparse LP (BASELINE config 4 by default) on the sparse-basis path,
with progress lines and an optimality certificate that needs no factorisation on the host: the
engine's final x gives primal feasibility (A x_S + slack = b, x >= 0, scipy sparse product), the
reduced costs of the slack variables ARE the dual vector (y_i = z of slack i when it is nonbasic,
0 when it is basic), so dual feasibility (A^T y >= c, y >= 0) and the duality gap b.y - c.x are
checked independently of the engine's basis inverse (strong duality).

  python3 tools/full_solve_sparse.py [rows cols per_col seed] [max seconds (1000)] [chunk (100000)]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from dantzig_amd import core

m, ns, per_col, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (50000, 100000, 50, 1004)
max_s = float(sys.argv[5]) if len(sys.argv) > 5 else 1000.0
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 100000
cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
t0 = time.time()
with core.Solver(lp, numerics=core.FAST, poll_interval=100, log_capacity=1) as s:
    status = "iter_limit"
    while status == "iter_limit" and time.time() - t0 < max_s:
        t1 = time.time()
        status = s.run(chunk)
        r = s.result(log=False)
        print(f"  {r.iterations:9d} pivots  {time.time() - t0:6.0f} s  {status:10s} k={r.dense_columns:6d}  "
              f"{chunk / (time.time() - t1):7.0f} it/s  objective {r.objective!r}  max_pivot_error "
              f"{r.max_pivot_error:.1e}  near ties {r.near_ties}  refactors {r.refactors}", flush=True)
    r = s.result(log=False)
print(f"sparse {m}x{ns}, {per_col} per column, seed {seed}: {status} after {r.iterations} pivots in "
      f"{time.time() - t0:.0f} s, objective {r.objective!r}")
a = sp.csc_matrix((val, ri, cp), shape=(m, ns))
xs, slack = np.zeros(ns), np.zeros(m)
for pos, var in enumerate(r.basis):
    if var < ns:
        xs[var] = r.x[pos]
    else:
        slack[var - ns] = r.x[pos]
y = np.zeros(m)
for pos, var in enumerate(r.nonbasis):
    if var >= ns:
        y[var - ns] = r.z[pos]
primal_res = float(np.abs(a @ xs + slack - b).max())
primal_neg = float(max(0.0, -xs.min(), -slack.min()))
dual_slack = a.T @ y - c
dual_neg = float(max(0.0, -dual_slack.min(), -y.min()))
pobj, dobj = float(c @ xs), float(b @ y)
print(f"certificate: |A x + s - b|max {primal_res:.2e}, most negative x/s {primal_neg:.2e}, most negative "
      f"dual slack / y {dual_neg:.2e}, c.x {pobj!r}, b.y {dobj!r}, gap (rel) "
      f"{abs(pobj - dobj) / max(1.0, abs(pobj)):.2e}, engine objective vs c.x (rel) "
      f"{abs(r.objective - pobj) / max(1.0, abs(pobj)):.2e}")

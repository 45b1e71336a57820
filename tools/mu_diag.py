"""FAST vs a committed oracle pivot log: where and by how much mu = -x/xbar (or -z/zbar) drifts.
  python3 tools/mu_diag.py tests/golden/oracle_pivots_1002_1024x2048.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
fx = np.load(sys.argv[1])
seed, m, ns = int(fx["seed"]), int(fx["m"]), int(fx["n_struct"])
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
for refi in (0, 500):
    got = core.solve(lp, numerics=core.FAST, poll_interval=64, refactor_interval=refi)
    mu = np.array([p[3] for p in got.pivots]); ref = fx["mu"]
    same = [(p[0], p[1], p[2]) for p in got.pivots] == list(zip(fx["kind"].tolist(), fx["entering"].tolist(), fx["leaving"].tolist()))
    d = np.abs(mu - ref); i = int(d.argmax())
    print(f"refactor_interval {refi}: pivots identical {same}; max |dmu| {d.max():.3e} at pivot {i} (mu {ref[i]:.6g}); "
          f"max |dmu|/max(1,|mu|) {(d / np.maximum(1, np.abs(ref))).max():.3e}; max_pivot_error {got.max_pivot_error:.2e}; "
          f"objective diff {got.objective - float(fx['objective']):.3e}")
    print("  quantiles of |dmu| (50, 90, 99, 99.9 %):", np.quantile(d, [0.5, 0.9, 0.99, 0.999]))

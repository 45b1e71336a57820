import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dantzig_amd import core
fx = np.load("/root/repo/tests/golden/oracle_pivots_2001_512x1024.npz")
a, b, c = core.gen_dense_lp(seed=2001, m=512, n_struct=1024)
lp = core.CoreLP.from_inequality_form(a, b, c)
for refi in (0, 200):
    got = core.solve(lp, numerics=core.FAST, poll_interval=64, refactor_interval=refi)
    mu = np.array([p[3] for p in got.pivots]); ref = fx["mu"]
    d = np.abs(mu - ref)
    i = int(d.argmax())
    print("refactor_interval", refi, "max abs diff", d.max(), "at pivot", i, "mu", ref[i], "rel", d[i]/abs(ref[i]),
          "max rel", (d/np.abs(ref)).max(), "max_pivot_error", got.max_pivot_error, "objective diff", got.objective - float(fx["objective"]))
    print("  quantiles of abs diff", np.quantile(d, [0.5, 0.9, 0.99, 0.999]))

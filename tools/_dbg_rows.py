import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
from tests.lp_families import make_lp
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9323
a, b, c = make_lp(seed, seed % 3, 2, 70)
print("shape", a.shape)
lp = core.CoreLP.from_inequality_form(a, b, c)
m = a.shape[0]; ns = a.shape[1]
print("m", lp.m, "n", lp.n, "T", int(0.9 * lp.m * (lp.n - lp.m) / (lp.n)))
for mode in ("rows", "columns"):
    os.environ.pop("DZG_PRICE_ROWS", None)
    if mode == "columns":
        os.environ["DZG_PRICE_ROWS"] = "0"
    with core.Solver(lp, numerics=core.FAST, poll_interval=1, seven_launches=int(os.environ.get("SEVEN", "0"))) as s:
        prev = 0.0
        for it in range(200):
            st = s.run(1)
            r = s.result()
            if r.max_pivot_error > max(prev, 1e-9):
                p = r.pivots[-1] if r.pivots else None
                print(mode, "iter", r.iterations, "err", r.max_pivot_error, "k", r.dense_columns, "pivot", p, "refactors", r.refactors)
                prev = r.max_pivot_error
            if st != "iter_limit":
                break
        print(mode, "end", st, r.iterations, r.max_pivot_error, r.dense_columns)
        print([ (i, p[0], p[1], p[2]) for i, p in enumerate(r.pivots[:40])])

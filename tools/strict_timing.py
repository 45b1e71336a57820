import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
for m, ns, seed in ((16, 32, 1), (64, 128, 4), (128, 256, 6), (192, 384, 7)):
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    t = time.perf_counter(); r = core.solve(lp, numerics=core.STRICT, log=False); dt = time.perf_counter() - t
    print(f"STRICT {m}x{ns}: {r.status} {r.iterations} pivots in {dt:.2f}s = {1e3*dt/max(r.iterations,1):.2f} ms/pivot", flush=True)

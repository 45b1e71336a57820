"""STRICT (bit-exact) per-pivot cost versus m, with the kernel classes that make it up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core, _ffi
for m, ns, seed, pivots in ((256, 512, 15, 40), (512, 1024, 2001, 30), (1024, 2048, 1002, 20),
                            (2048, 4096, 2002, 10), (4096, 8192, 1006, 4)):
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.STRICT) as s:
        s.run(2)
        t = time.perf_counter(); s.run(pivots); dt = time.perf_counter() - t
    print(f"STRICT {m}x{ns}: {1e3*dt/pivots:.2f} ms/pivot", flush=True)

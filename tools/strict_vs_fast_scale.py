"""One-off parity evidence at the benchmark size: the first pivots of the 8192x16384 LP in STRICT
(reference arithmetic on the GPU, bit-identical to the oracle at every size both can run) and in
FAST numerics.  STRICT needs ~2.7 s per pivot here, so it runs in chunks with progress lines.
Output is committed under profiles/.

  python3 tools/strict_vs_fast_scale.py [pivots] [rows] [cols] [seed] [nonzeros per column: CSC input]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core

pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1003
per_col = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if per_col > 0:  # generator G2, matrix kept CSC on the device
    cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
    lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
else:
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
t = time.time()
with core.Solver(lp, numerics=core.STRICT, max_iter=pivots) as s:
    status, done = "iter_limit", 0
    while status == "iter_limit" and done < pivots:
        status = s.run(min(20, pivots - done))
        done = s.result(log=False).iterations
        print(f"  STRICT {done} pivots, {time.time() - t:.0f} s", flush=True)
    strict = s.result()
ts = time.time() - t
t = time.time(); fast = core.solve(lp, numerics=core.FAST, max_iter=pivots); tf = time.time() - t
ls = [(k, e, l) for k, e, l, _ in strict.pivots]; lf = [(k, e, l) for k, e, l, _ in fast.pivots]
print(f"{m}x{ns} seed {seed}{f' CSC {per_col} per column' if per_col else ''}: {pivots} pivots  STRICT {ts:.1f}s  FAST {tf:.2f}s")
print("pivot logs identical:", ls == lf)
if ls != lf:
    first = next(i for i, (p, q) in enumerate(zip(ls, lf)) if p != q)
    print("first difference at pivot", first, "strict", ls[first], "fast", lf[first])
n = min(len(strict.pivots), len(fast.pivots))
print("max |mu_fast - mu_strict| / |mu|:", max(abs(p[3] - q[3]) / abs(q[3]) for p, q in zip(fast.pivots[:n], strict.pivots[:n])))
print("max |x_fast - x_strict|:", float(np.abs(fast.x - strict.x).max()))
print("FAST max_pivot_error:", fast.max_pivot_error)
import hashlib
print("sha256 of the STRICT pivot log:", hashlib.sha256(repr(ls).encode()).hexdigest())
print("log:", ls)

"""STRICT numerics (the reference's arithmetic on the GPU) against a first-pivots fixture of the CPU
oracle at a size where the oracle needs seconds to minutes per pivot
(tests/golden/oracle_{first,blocked}_pivots_<seed>_<m>x<ns>.json): kind, entering, leaving of every
pivot and every bit of mu.  FAST is run over the same pivots beside it.

  python3 tools/strict_replay_first_pivots.py [first|blocked] [seed m ns] [pivots (all)]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "blocked"
seed, m, ns = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1003, 8192, 16384)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                    f"oracle_{kind}_pivots_{seed}_{m}x{ns}.json")
fx = json.load(open(path))
n = int(sys.argv[5]) if len(sys.argv) > 5 else len(fx["kind"])
n = min(n, len(fx["kind"]))
want = list(zip(fx["kind"][:n], fx["entering"][:n], fx["leaving"][:n]))
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
for name, numerics in (("FAST", core.FAST), ("STRICT", core.STRICT)):
    t0 = time.perf_counter()
    r = core.solve(lp, numerics=numerics, max_iter=n)
    dt = time.perf_counter() - t0
    got = [(k, e, l) for k, e, l, _ in r.pivots]
    mu = np.array([p[3] for p in r.pivots])
    ref = np.array(fx["mu"][:n])
    same_log = got == want
    first_diff = next((i for i, (g, w) in enumerate(zip(got, want)) if g != w), None)
    bits = int(np.count_nonzero(mu.view(np.uint64) != ref.view(np.uint64)))
    rel = float(np.max(np.abs(mu - ref) / np.abs(ref)))
    print(f"{name:6s} {m}x{ns} seed {seed}: {len(got)} pivots in {dt:.1f} s ({dt / max(len(got), 1):.3f} s per pivot); "
          f"pivot log == oracle's ({os.path.basename(path)}): {same_log}"
          + ("" if same_log else f" (first difference at pivot {first_diff})")
          + f"; mu: {bits} of {n} differ in some bit, largest relative difference {rel:.2e}; "
          f"near ties {r.near_ties}", flush=True)

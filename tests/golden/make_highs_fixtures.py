"""Optimal objective values of seeded G1 LPs from an INDEPENDENT solver (scipy / HiGHS),
written to tests/golden/highs_objectives.json.  They check the objective parity of the GPU
solves (1e-9 relative, BASELINE.json north_star) at sizes the CPU oracle cannot finish;
they say nothing about pivot order.   Run: python tests/golden/make_highs_fixtures.py"""
import json
import os
import sys
import time

import numpy as np
from scipy.optimize import linprog

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dantzig_amd import core  # noqa: E402  (host-side generator only, no GPU needed)

CASES = [(1002, 1024, 2048), (2001, 512, 1024), (2002, 2048, 4096), (2003, 300, 900)]

if __name__ == "__main__":
    out = []
    for seed, m, ns in CASES:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        t = time.time()
        r = linprog(-c, A_ub=np.array(a), b_ub=b, bounds=(0, None), method="highs")
        assert r.status == 0, r.message
        out.append({"seed": seed, "m": m, "n_struct": ns, "objective": float(-r.fun),
                    "solver": "scipy.optimize.linprog(method='highs')", "seconds": round(time.time() - t, 1)})
        print(out[-1], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "highs_objectives.json"), "w") as f:
        json.dump(out, f, indent=1)

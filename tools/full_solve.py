"""Solve a whole G1 LP with the FAST engine and certify the result independently on the host:
given only the final basis, LAPACK (numpy) recomputes x_B = B^-1 b and y = B^-T c_B and checks
primal feasibility, dual feasibility and the duality gap -- an optimality certificate that does
not depend on the engine's arithmetic (strong duality), usable at sizes no CPU simplex reaches.

  python3 tools/full_solve.py [rows] [cols] [seed] [refactor_interval]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core  # noqa: E402
from tests.optimality import certificate  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1003
    refi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(lp, numerics=core.FAST, poll_interval=256, refactor_interval=refi,
                     log_capacity=1) as s:
        t0 = time.perf_counter()
        status = "iter_limit"
        last = t0
        while status == "iter_limit":
            status = s.run(20000)
            now = time.perf_counter()
            r = s.result(log=False)
            print(f"  {r.iterations} pivots, {now - t0:.1f} s, status {status}, objective "
                  f"{r.objective!r}, max_pivot_error {r.max_pivot_error:.2e}", flush=True)
            last = now
        dt = last - t0
        res = s.result(log=False)
    print(f"{rows}x{cols} seed {seed}: {status} after {res.iterations} pivots in {dt:.2f} s "
          f"({res.iterations / dt:.0f} it/s), objective {res.objective!r}")
    t0 = time.perf_counter()
    cert = certificate(np.asarray(a), b, c, res.basis)
    gap = abs(cert["primal_obj"] - cert["dual_obj"]) / max(1.0, abs(cert["primal_obj"]))
    rel = abs(res.objective - cert["primal_obj"]) / max(1.0, abs(cert["primal_obj"]))
    print(f"host certificate ({time.perf_counter() - t0:.1f} s): {cert}")
    print(f"duality gap (rel) {gap:.2e}; engine objective vs LAPACK basic solution (rel) {rel:.2e}")


if __name__ == "__main__":
    main()

"""Complete pivot logs of seeded G1 LPs under the CPU oracle (the C restatement of the
reference's arithmetic, oracle/), written to tests/golden/oracle_pivots_<seed>_<m>x<ns>.npz.

The oracle needs about 0.1 s per pivot at 1024 rows (two dense LUs per iteration, as the
reference), so whole solves at BASELINE.json's config 2 take the better part of an hour of CPU:
they are computed once here and committed, and the GPU tests replay them
(tests/test_gpu_parity.py::test_whole_solve_follows_the_oracle_pivot_log).

Run: python tests/golden/make_oracle_pivot_logs.py [seed m ns]...
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dantzig_amd import core  # noqa: E402  (host-side generator only, no GPU needed)
from oracle import oracle as ora  # noqa: E402

CASES = [(2001, 512, 1024), (1002, 1024, 2048)]

if __name__ == "__main__":
    argv = [int(v) for v in sys.argv[1:]]
    cases = [tuple(argv[i:i + 3]) for i in range(0, len(argv), 3)] or CASES
    for seed, m, ns in cases:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        t = time.time()
        res = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=4_000_000)
        dt = time.time() - t
        path = os.path.join(ROOT, "tests", "golden", f"oracle_pivots_{seed}_{m}x{ns}.npz")
        np.savez_compressed(
            path, seed=seed, m=m, n_struct=ns, status=res.status, iterations=res.iterations,
            objective=res.objective, oracle_seconds=round(dt, 1),
            kind=np.array([p[0] for p in res.pivots], dtype=np.int8),
            entering=np.array([p[1] for p in res.pivots], dtype=np.int32),
            leaving=np.array([p[2] for p in res.pivots], dtype=np.int32),
            mu=np.array([p[3] for p in res.pivots], dtype=np.float64),
            basis=np.asarray(res.basis, dtype=np.int32))
        print(f"{path}: {res.status} after {res.iterations} pivots, objective {res.objective!r}, "
              f"{dt:.0f} s", flush=True)

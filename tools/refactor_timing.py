"""Cost of dzg_solver_refactor -- the on-device rebuild of the basis inverse: blocked LU with
partial pivoting of the k x k structural block (panels on the vector ALUs, trailing updates and
the block substitutions on the fp64 matrix cores), X = G^-1, then the basic-slack rows of the
inverse, one more MFMA GEMM -- at chosen k, on a warm-started 8192-row G1 LP whose basis holds
exactly k structural columns (any k dense random columns are a nonsingular block).

  python3 tools/refactor_timing.py [rows (8192)] [k values, comma separated (2048,4096,8192)]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ks = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "2048,4096,8192").split(",")]
ns = 2 * m
a, b, c = core.gen_dense_lp(seed=1003, m=m, n_struct=ns)
cc = np.concatenate([c, np.zeros(m)])
for k in ks:
    # basis: structural columns 0..k-1 at positions 0..k-1, slacks of rows k..m-1 behind them
    basis = np.concatenate([np.arange(k), ns + np.arange(k, m)]).astype(np.int64)
    nonbasis = np.concatenate([np.arange(k, ns), ns + np.arange(k)]).astype(np.int64)
    lp = core.CoreLP(a=np.asarray(a), c=cc, basis=basis, nonbasis=nonbasis, x=np.ones(m),
                     z=-np.ones(ns))  # (x > 0 > z: pivots to make on the fresh inverse)
    t0 = time.perf_counter()
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1) as s:
        t_create = time.perf_counter() - t0   # upload + the initial refactorisation
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            s.refactor()
            times.append(time.perf_counter() - t0)
        # the fresh inverse at work: 64 pivots on it, and the health monitor's figure for them (the
        # pivot element by FTRAN against the same element by BTRAN + pricing)
        st = s.run(64)
        r = s.result(log=False)
        check = f"{r.iterations} pivots on the fresh inverse: {st}, max_pivot_error {r.max_pivot_error:.1e}"
    dt = min(times)
    lu = (2.0 / 3.0) * k ** 3           # LU of G
    inv = (4.0 / 3.0) * k ** 3          # forward + backward substitution of the identity
    rows = 2.0 * (m - k) * k * k        # basic-slack rows: (m - k) x k x k GEMM
    print(f"m={m} k={k}: refactor {dt * 1e3:8.1f} ms (3 runs: {[round(t * 1e3, 1) for t in times]}), "
          f"{(lu + inv + rows) / dt / 1e12:6.2f} TFLOP/s over {(lu + inv + rows) / 1e9:.0f} GFLOP "
          f"(LU {lu / 1e9:.0f} + inverse {inv / 1e9:.0f} + slack rows {rows / 1e9:.0f}); "
          f"refactors {r.refactors}, create {t_create:.2f} s; {check}", flush=True)

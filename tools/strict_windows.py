"""FAST against STRICT deep inside a whole solve ("parity reach at size"): windows of W
pivots at chosen pivot counts of the benchmark LP.  FAST runs the solve from the slack basis; at each
window start its state (basis, nonbasis, x, xbar, z, zbar: core.resumed_from) is handed to a STRICT
solver -- the reference's arithmetic on the GPU: a dense LU of B and of B^T per pivot -- which takes W
pivots from there, while FAST simply carries on.  The two logs must agree pivot for pivot; mu is
compared to its relative difference.  (STRICT from FAST's state, not from pivot 0: 0.6 s per pivot at
8192 rows puts pivot 400 000 three days away.)

  python3 tools/strict_windows.py [rows cols seed] [window] [start pivots ...] [--sparse-per-col N]

--sparse-per-col N: generator G2 (BASELINE config 4: N nonzeros per column, matrix CSC on the device, FAST on
the sparse-basis path; STRICT densifies the basis like the reference does).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from dantzig_amd import core  # noqa: E402

per_col = 0
if "--sparse-per-col" in sys.argv:
    i = sys.argv.index("--sparse-per-col")
    per_col = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
m, ns, seed = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 16384, 1003)
window = int(sys.argv[4]) if len(sys.argv) > 4 else 40
starts = [int(v) for v in sys.argv[5:]] or [20000, 150000, 400000]
if per_col > 0:
    cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
    lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
else:
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
print(f"{m}x{ns} seed {seed}{' (%d nonzeros per column, CSC)' % per_col if per_col else ''}: windows of {window} pivots at {starts}", flush=True)
with core.Solver(lp, numerics=core.FAST, poll_interval=50, log_capacity=max(starts) + window + 64) as s:
    done = 0
    for start in starts:
        t = time.time()
        status = s.run(start - done)
        r0 = s.result(log=False)
        done = r0.iterations
        if status != "iter_limit" or done != start:
            print(f"  FAST stopped with {status} after {done} pivots", flush=True)
            break
        status = s.run(window)
        r1 = s.result()
        done = r1.iterations
        fast = r1.pivots[start:start + window]
        print(f"  pivot {start}: FAST there in {time.time() - t:.1f} s, k = {r0.dense_columns}, near ties so far "
              f"{r0.near_ties}, max_pivot_error {r0.max_pivot_error:.2e}", flush=True)
        t = time.time()
        strict = core.solve(core.resumed_from(lp, r0), numerics=core.STRICT, max_iter=len(fast))
        ts = time.time() - t
        same = [(k, e, l) for k, e, l, _ in strict.pivots] == [(k, e, l) for k, e, l, _ in fast]
        n = min(len(strict.pivots), len(fast))
        rel = max((abs(strict.pivots[i][3] - fast[i][3]) / max(abs(strict.pivots[i][3]), 1e-300) for i in range(n)),
                  default=0.0)
        first_diff = next((i for i in range(n) if strict.pivots[i][:3] != fast[i][:3]), -1)
        print(f"  pivot {start}: STRICT took {len(strict.pivots)} pivots from FAST's state in {ts:.1f} s: "
              f"{'the same pivots' if same else 'DIFFERENT at window pivot %d' % first_diff}, "
              f"largest relative difference of mu {rel:.2e}", flush=True)

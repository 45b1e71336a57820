"""Fuzz the GPU engine against the CPU oracle on many small LPs: continuous G1 data and
small-integer data (exact ties in both pivot rules, degenerate vertices, unbounded / infeasible
outcomes under the reference's one-sided status() quirk).  STRICT must reproduce the oracle bit
for bit; FAST is expected to take the same pivots (reported, not required, on integer data).

  python3 tools/fuzz_parity.py [cases] [first_seed] [max rows (default 70)] [iteration cap (default 20000)]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core
from oracle import oracle as ora

def same_bits(x, y):
    """Bitwise equal, except that zeros of either sign and NaNs of any payload match."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    return x.shape == y.shape and bool(np.all((x.view(np.int64) == y.view(np.int64))
                                              | ((x == 0) & (y == 0)) | (np.isnan(x) & np.isnan(y))))


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_m = int(sys.argv[3]) if len(sys.argv) > 3 else 70
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
bad_strict, bad_fast, statuses = [], [], {}
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    m, ns = int(rng.integers(1, max_m)), int(rng.integers(1, 2 * max_m))
    kind = case % 3
    if kind == 0:
        a, b, c = core.gen_dense_lp(seed=seed0 + case, m=m, n_struct=ns)
        a = np.array(a)
    elif kind == 1:  # small integers, many zeros: ties everywhere
        a = rng.integers(-3, 4, (m, ns)).astype(np.float64)
        b = rng.integers(-2, 9, m).astype(np.float64)
        c = rng.integers(-4, 5, ns).astype(np.float64)
    else:            # 0/1 matrix, nonnegative rhs: degenerate primal vertices
        a = (rng.uniform(size=(m, ns)) < 0.3).astype(np.float64)
        b = rng.integers(0, 4, m).astype(np.float64)
        c = rng.integers(-1, 6, ns).astype(np.float64)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=cap)
    statuses[want.status] = statuses.get(want.status, 0) + 1
    lp = core.CoreLP.from_inequality_form(a, b, c)
    wlog = [(k, e, l) for k, e, l, _ in want.pivots]
    s = core.solve(lp, numerics=core.STRICT, max_iter=cap)
    ok = (s.status == want.status and [(k, e, l) for k, e, l, _ in s.pivots] == wlog
          and same_bits([p[3] for p in s.pivots], [p[3] for p in want.pivots])
          and all(same_bits(getattr(s, f), getattr(want, f)) for f in ("x", "xbar", "z", "zbar")))
    if not ok:
        bad_strict.append((seed0 + case, kind, m, ns, s.status, want.status))
    f = core.solve(lp, numerics=core.FAST, max_iter=cap, poll_interval=8)
    if f.status != want.status or [(k, e, l) for k, e, l, _ in f.pivots] != wlog:
        n_same = next((i for i, (p, q) in enumerate(zip([(k, e, l) for k, e, l, _ in f.pivots], wlog)) if p != q),
                      min(len(f.pivots), len(wlog)))
        bad_fast.append((seed0 + case, kind, m, ns, f.status, want.status, n_same, len(wlog)))
    if (case + 1) % 25 == 0:
        print(f"  {case + 1} cases, {time.time() - t0:.0f} s; STRICT mismatches {len(bad_strict)}, "
              f"FAST mismatches {len(bad_fast)}", flush=True)
print(f"{cases} cases from seed {seed0}: oracle outcomes {statuses}")
print("STRICT mismatches (seed, kind, m, ns, got, want):", bad_strict)
print("FAST mismatches (seed, kind, m, ns, got, want, identical pivots, oracle pivots):", bad_fast)

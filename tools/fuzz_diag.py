"""Which component of a STRICT result differs from the oracle for given fuzz seeds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core
from oracle import oracle as ora

def make(case, kind=None):
    rng = np.random.default_rng(case)
    m, ns = int(rng.integers(1, 70)), int(rng.integers(1, 140))
    kind = case % 3 if kind is None else kind
    if kind == 0:
        a, b, c = core.gen_dense_lp(seed=case, m=m, n_struct=ns); a = np.array(a)
    elif kind == 1:
        a = rng.integers(-3, 4, (m, ns)).astype(np.float64); b = rng.integers(-2, 9, m).astype(np.float64); c = rng.integers(-4, 5, ns).astype(np.float64)
    else:
        a = (rng.uniform(size=(m, ns)) < 0.3).astype(np.float64); b = rng.integers(0, 4, m).astype(np.float64); c = rng.integers(-1, 6, ns).astype(np.float64)
    return a, b, c

def same_bits(x, y):
    x, y = np.asarray(x, float), np.asarray(y, float)
    return bool(np.all((x.view(np.int64) == y.view(np.int64)) | ((x == 0) & (y == 0)) | (np.isnan(x) & np.isnan(y))))

for arg in sys.argv[1:]:  # "seed" or "seed:kind" (fuzz_parity's kind is (seed - first_seed) % 3)
    case, kind = (int(v) for v in arg.split(":")) if ":" in arg else (int(arg), None)
    a, b, c = make(case, kind)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=20000)
    s = core.solve(core.CoreLP.from_inequality_form(a, b, c), numerics=core.STRICT, max_iter=20000)
    wl = [(k, e, l) for k, e, l, _ in want.pivots]; sl = [(k, e, l) for k, e, l, _ in s.pivots]
    n = min(len(wl), len(sl))
    first = next((i for i in range(n) if wl[i] != sl[i]), n)
    mu_ok = same_bits([p[3] for p in s.pivots[:n]], [p[3] for p in want.pivots[:n]])
    mus, muw = [p[3] for p in s.pivots[:n]], [p[3] for p in want.pivots[:n]]
    bad_mu = [i for i in range(n) if not same_bits([mus[i]], [muw[i]])]
    if bad_mu:
        i = bad_mu[0]
        print(f"   first differing mu at pivot {i}: strict {mus[i]!r} oracle {muw[i]!r}")
    for name in ("x", "xbar", "z", "zbar"):
        g, w = getattr(s, name), getattr(want, name)
        bad = [i for i in range(len(w)) if not same_bits([g[i]], [w[i]])]
        if bad:
            print(f"   {name}: {len(bad)} entries differ, first at {bad[0]}: strict {g[bad[0]]!r} oracle {w[bad[0]]!r}")
    print(f"seed {case}: status {s.status}/{want.status} pivots {len(sl)}/{len(wl)} identical prefix {first}; mu bits {mu_ok}; "
          f"x {same_bits(s.x, want.x)} xbar {same_bits(s.xbar, want.xbar)} z {same_bits(s.z, want.z)} zbar {same_bits(s.zbar, want.zbar)}; "
          f"nan in oracle x/z: {int(np.isnan(want.x).sum())}/{int(np.isnan(want.z).sum())}")
    if first < n:
        print("   first differing pivot: strict", sl[first], "oracle", wl[first])

"""Seeded LP families shared by the parity tests, the golden-fixture scripts and tools/fuzz_parity.py.

kind 0  continuous G1 data -- near ties are measure-zero
kind 1  small integers, many zeros -- exact ties in both pivot rules, 0/0 and x/0 ratios
kind 2  0/1 matrix, nonnegative integer rhs -- degenerate primal vertices
All are  max c.x  st  A x <= b, x >= 0  entered at the core boundary (slack basis).
"""
import hashlib

import numpy as np


def make_lp(seed: int, kind: int, min_m: int, max_m: int):
    rng = np.random.default_rng(seed)
    m, ns = int(rng.integers(min_m, max_m)), int(rng.integers(min_m, 2 * max_m))
    if kind == 0:
        from dantzig_amd import core

        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        a = np.array(a)
    elif kind == 1:
        a = rng.integers(-3, 4, (m, ns)).astype(np.float64)
        b = rng.integers(-2, 9, m).astype(np.float64)
        c = rng.integers(-4, 5, ns).astype(np.float64)
    else:
        a = (rng.uniform(size=(m, ns)) < 0.3).astype(np.float64)
        b = rng.integers(0, 4, m).astype(np.float64)
        c = rng.integers(-1, 6, ns).astype(np.float64)
    return a, b, c


def log3(pivots):
    """(kind, entering, leaving) per pivot."""
    return [(int(p[0]), int(p[1]), int(p[2])) for p in pivots]


def log_digest(pivots) -> str:
    """sha256 over the (kind, entering, leaving) triples of a pivot log."""
    arr = np.array(log3(pivots), dtype=np.int64).reshape(-1, 3)
    return hashlib.sha256(arr.tobytes()).hexdigest()

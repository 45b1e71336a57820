"""Fuzz the GPU engine against the CPU oracle on many small LPs: continuous G1 data,
small-integer data (exact ties in both pivot rules, degenerate vertices, unbounded / infeasible
outcomes under the reference's one-sided status() quirk) and 0/1 data.

  STRICT  must reproduce the oracle bit for bit (pivot log, mu, x, xbar, z, zbar).
  FAST    may leave the oracle's path only at a pivot it has FLAGGED as a near tie
          (dzg_result.first_near_tie <= first differing pivot): "unflagged divergences" must be 0.
  AUTO    (dzg_core_solve: FAST that stops at the first near tie, then STRICT from the first
          pivot) must equal the oracle in status and pivot log: "another verdict" must be 0.

  python3 tools/fuzz_parity.py [cases] [first_seed] [max rows (70)] [iteration cap (20000)]
                               [min rows (1)] [families, e.g. 12 = integer + 0/1 only] [csc]

With a trailing `csc` the same LPs are handed over as CSC (zeros dropped, like the reference's
CscMatrix): FAST then runs on the sparse-basis path (k_sparse.hip), STRICT on its CSC gathers.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core
from oracle import oracle as ora
from tests.lp_families import log3, make_lp


def same_bits(x, y):
    """Bitwise equal, except that zeros of either sign and NaNs of any payload match."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    return x.shape == y.shape and bool(np.all((x.view(np.int64) == y.view(np.int64))
                                              | ((x == 0) & (y == 0)) | (np.isnan(x) & np.isnan(y))))


def first_difference(p, q):
    return next((i for i, (a, b) in enumerate(zip(p, q)) if a != b), min(len(p), len(q)))


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    max_m = int(sys.argv[3]) if len(sys.argv) > 3 else 70
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
    min_m = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    families = [int(ch) for ch in sys.argv[6]] if len(sys.argv) > 6 else [0, 1, 2]
    as_csc = "csc" in sys.argv[7:]
    bad_strict, bad_auto, unflagged, flagged_div, statuses = [], [], [], [], {}
    false_pos = {0: 0, 1: 0, 2: 0}
    clean = {0: 0, 1: 0, 2: 0}
    per_kind = {0: 0, 1: 0, 2: 0}
    auto_strict = 0
    t0 = time.time()
    for case in range(cases):
        seed, kind = seed0 + case, families[case % len(families)]
        a, b, c = make_lp(seed, kind, min_m, max_m)
        m, ns = a.shape
        per_kind[kind] += 1
        want = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=cap)
        statuses[want.status] = statuses.get(want.status, 0) + 1
        if as_csc:
            import scipy.sparse as sp
            acsc = sp.csc_matrix(a)
            acsc.eliminate_zeros()
            acsc.sort_indices()
            lp = core.CoreLP.from_csc(m, acsc.indptr, acsc.indices, acsc.data, b, c)
        else:
            lp = core.CoreLP.from_inequality_form(a, b, c)
        wlog = log3(want.pivots)
        s = core.solve(lp, numerics=core.STRICT, max_iter=cap)
        ok = (s.status == want.status and log3(s.pivots) == wlog
              and same_bits([p[3] for p in s.pivots], [p[3] for p in want.pivots])
              and all(same_bits(getattr(s, f), getattr(want, f)) for f in ("x", "xbar", "z", "zbar")))
        if not ok:
            bad_strict.append((seed, kind, m, ns, s.status, want.status))
        # FAST, near ties counted: a divergence must have been flagged at or before it happens
        f = core.solve(lp, numerics=core.FAST, max_iter=cap, poll_interval=8)
        diverged = f.status != want.status or log3(f.pivots) != wlog
        if diverged:
            d = first_difference(log3(f.pivots), wlog)
            row = (seed, kind, m, ns, f.status, want.status, d, len(wlog), f.first_near_tie,
                   f"{f.margins[:d + 1].min() if d < len(f.margins) else f.min_margin:.2e}",
                   f"{f.max_pivot_error:.1e}")
            if 0 <= f.first_near_tie <= d:
                flagged_div.append(row)
            else:
                unflagged.append(row)
        elif f.near_ties > 0:
            false_pos[kind] += 1
        else:
            clean[kind] += 1
        # AUTO through dzg_core_solve; small LPs would be STRICT by size, so force the FAST branch
        if m > 2:
            r = core.core_solve(lp, numerics=core.AUTO, max_iter=cap, auto_strict_rows=2,
                                poll_interval=8, log_cap=max(cap, 1))
            auto_strict += r.numerics == "strict"
            if r.status != want.status or log3(r.pivots) != wlog:
                bad_auto.append((seed, kind, m, ns, r.status, want.status, r.numerics,
                                 first_difference(log3(r.pivots), wlog), len(wlog)))
        if (case + 1) % 25 == 0:
            print(f"  {case + 1} cases, {time.time() - t0:.0f} s; STRICT mismatches {len(bad_strict)}, "
                  f"FAST divergences flagged {len(flagged_div)} / UNFLAGGED {len(unflagged)}, "
                  f"AUTO mismatches {len(bad_auto)}", flush=True)
    print(f"{cases} cases from seed {seed0}{' (CSC input)' if as_csc else ''}, rows {min_m}..{max_m - 1}, families {families} "
          f"(0 continuous G1, 1 small integers, 2 zero/one): oracle outcomes {statuses}")
    print("STRICT mismatches (seed, kind, m, ns, got, want):", bad_strict)
    print(f"FAST (near ties counted): per family cases {per_kind}; followed the oracle unflagged "
          f"{clean}; followed it but flagged some pivot (false positives) {false_pos}")
    print(f"FAST divergences flagged at or before the differing pivot: {len(flagged_div)}")
    print("FAST UNFLAGGED divergences (seed, kind, m, ns, got, want, first differing pivot, oracle "
          "pivots, first_near_tie, min margin up to there, max_pivot_error):", unflagged)
    print(f"AUTO (dzg_core_solve): {auto_strict} re-solved in STRICT; mismatches (seed, kind, m, "
          f"ns, got, want, numerics, first differing pivot, oracle pivots):", bad_auto)
    if "-v" in sys.argv:
        for row in flagged_div:
            print("  flagged:", row)

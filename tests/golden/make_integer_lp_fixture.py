"""Writes tests/golden/integer_lps_200_400.json: the CPU oracle's outcome on 200 seeded
small-integer and 0/1 LPs of 200-400 rows (tests/lp_families.py, kinds 1 and 2), each run for at
most CAP pivots: status, pivot count and the sha256 of the pivot log.  These are the LPs on which
FAST numerics used to leave the reference's path silently (exact ties, 0/0 and x/0 ratios); the
GPU suite checks that AUTO numerics reproduces every one of them
(tests/test_gpu_parity.py::test_auto_follows_the_oracle_on_integer_lps).

  python3 tests/golden/make_integer_lp_fixture.py        (about 3 minutes on 6 cores)
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FIRST_SEED, CASES, CAP, MIN_M, MAX_M = 9000, 200, 120, 200, 401


def one(seed):
    from oracle import oracle as ora
    from tests.lp_families import log_digest, make_lp

    kind = 1 + seed % 2
    a, b, c = make_lp(seed, kind, MIN_M, MAX_M)
    r = ora.simplex_solve(ora.stdform_from_dense(a, b, c), max_iter=CAP)
    return {"seed": seed, "kind": kind, "m": int(a.shape[0]), "ns": int(a.shape[1]),
            "status": r.status, "pivots": int(r.iterations), "sha256": log_digest(r.pivots)}


if __name__ == "__main__":
    with ProcessPoolExecutor(max_workers=6) as ex:
        rows = list(ex.map(one, range(FIRST_SEED, FIRST_SEED + CASES)))
    out = {"cap": CAP, "min_m": MIN_M, "max_m": MAX_M, "cases": rows}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "integer_lps_200_400.json"), "w") as f:
        json.dump(out, f, indent=0)
    tally = {}
    for r in rows:
        tally[r["status"]] = tally.get(r["status"], 0) + 1
    print(len(rows), "cases:", tally)

"""Writes tests/golden/pivots_1005_32768x65536.json: the first 80 pivots of BASELINE config 5
(dense 32768 x 65536, generator G1 seed 1005).

The arbiter at this size is STRICT numerics (the reference's arithmetic on the GPU, bit-identical
to the CPU oracle at every size both can run): round 1 ran it for these 80 pivots -- 1 013 s,
12.7 s per pivot -- and kept the sha256 of its log in
profiles/r01_strict_vs_fast_32768x65536_80pivots.txt, where FAST numerics produced the identical
log.  This script re-runs FAST (2 s), checks its log against THAT sha256 -- so what is written is
the STRICT log, pivot for pivot -- and stores it.  Needs an MI355X.

  python3 tests/golden/make_config5_pivot_fixture.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STRICT_SHA256 = "57337b934484f0a3598a00ec14cd93f9bb4455cb86a939ceb2a4ba71e2900106"
M, NS, SEED, PIVOTS = 32768, 65536, 1005, 80

if __name__ == "__main__":
    from dantzig_amd import core

    a, b, c = core.gen_dense_lp(seed=SEED, m=M, n_struct=NS)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    fast = core.solve(lp, numerics=core.FAST, max_iter=PIVOTS)
    log = [(k, e, l) for k, e, l, _ in fast.pivots]
    sha = hashlib.sha256(repr(log).encode()).hexdigest()
    print("near ties:", fast.near_ties, " min margin:", fast.min_margin, " sha256:", sha)
    if sha != STRICT_SHA256:
        sys.exit("FAST's log is not the STRICT log of round 1: nothing written")
    out = {"m": M, "n_struct": NS, "seed": SEED, "pivots": PIVOTS, "strict_sha256": STRICT_SHA256,
           "source": "profiles/r01_strict_vs_fast_32768x65536_80pivots.txt",
           "kind": [p[0] for p in log], "entering": [p[1] for p in log],
           "leaving": [p[2] for p in log], "mu_fast": [p[3] for p in fast.pivots]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pivots_1005_32768x65536.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)

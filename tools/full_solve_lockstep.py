"""Whole solve of a G1 LP through the column-sharded path (all P ranks in lockstep on one GPU),
with the host LAPACK optimality certificate.  Pivot count, basis and objective must be those of the
single-GPU solve (profiles/r01_full_solve_8192x16384.txt).

  python3 tools/full_solve_lockstep.py [P] [rows] [cols] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dantzig_amd import core, sharded
from tests.optimality import certificate

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1003
a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
lp = core.CoreLP.from_inequality_form(a, b, c)
solvers = sharded.make_lockstep(lp, P, poll_interval=256, log_capacity=1)
t0 = time.time()
status = "iter_limit"
while status == "iter_limit":
    status = sharded.run_lockstep(solvers, 50000)
    r = solvers[0].result(log=False)
    print(f"  {r.iterations} pivots, {time.time() - t0:.0f} s, {status}, objective {r.objective!r}, "
          f"max_pivot_error {r.max_pivot_error:.2e}", flush=True)
dt = time.time() - t0
results = [s.result(log=False) for s in solvers]
for s in solvers:
    s.close()
r0 = results[0]
print(f"{rows}x{cols} seed {seed}, {P} ranks in lockstep: {status} after {r0.iterations} pivots in {dt:.0f} s, "
      f"objective {r0.objective!r}")
print("all ranks: same basis", all(np.array_equal(r.basis, r0.basis) for r in results),
      "same x bits", all(np.array_equal(r.x.view(np.int64), r0.x.view(np.int64)) for r in results))
cert = certificate(np.asarray(a), b, c, r0.basis)
print("host certificate:", cert)
print("duality gap (rel)", abs(cert["primal_obj"] - cert["dual_obj"]) / max(1, abs(cert["primal_obj"])),
      "; engine objective vs LAPACK (rel)", abs(r0.objective - cert["primal_obj"]) / max(1, abs(cert["primal_obj"])))

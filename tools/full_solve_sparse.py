"""Whole FAST solve of a G2 sparse LP (BASELINE config 4 by default) on the sparse-basis path,
with progress lines and an optimality certificate that needs no factorisation on the host: the
engine's final x gives primal feasibility (A x_S + slack = b, x >= 0, scipy sparse product), the
reduced costs of the slack variables ARE the dual vector (y_i = z of slack i when it is nonbasic,
0 when it is basic), so dual feasibility (A^T y >= c, y >= 0) and the duality gap b.y - c.x are
checked independently of the engine's basis inverse (strong duality).

  python3 tools/full_solve_sparse.py [rows cols per_col seed] [max seconds (1000)] [chunk (100000)]
                                     [--state file.npz] [--state-out file.npz]

--state: a solve that outlasts one run is carried on in the next: the six state arrays of the last
result (basis, nonbasis, x, xbar, z, zbar -- a few MB; the k x k inverse is NOT saved) are written
there when the time is up, and read from there when the file exists: the new solver factorises
that basis on the device (dzg_lp.xbar / zbar) and goes on.  --state-out: where to write (default:
the --state file).  Under gpurun, gpurun_out/ comes back but does not travel out: write there, copy
the file to carry/ (git-ignored, travels) and read it from there in the next call.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from dantzig_amd import core

state_path = state_out = None
for flag in ("--state", "--state-out"):
    if flag in sys.argv:
        i = sys.argv.index(flag)
        if flag == "--state":
            state_path = sys.argv[i + 1]
        else:
            state_out = sys.argv[i + 1]
        del sys.argv[i:i + 2]
state_out = state_out or state_path
m, ns, per_col, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (50000, 100000, 50, 1004)
max_s = float(sys.argv[5]) if len(sys.argv) > 5 else 1000.0
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 100000
cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
done_before, seconds_before = 0, 0.0
if state_path and os.path.exists(state_path):
    st = np.load(state_path)
    assert (int(st["m"]), int(st["ns"]), int(st["per_col"]), int(st["seed"])) == (m, ns, per_col, seed)
    lp = core.resumed_from(lp, st)
    done_before, seconds_before = int(st["pivots"]), float(st["seconds"])
    print(f"resuming after {done_before} pivots ({seconds_before:.0f} s of earlier runs), "
          f"{int((st['basis'] < ns).sum())} structural columns in the basis", flush=True)
t0 = time.time()
with core.Solver(lp, numerics=core.FAST, poll_interval=100, log_capacity=1) as s:
    if done_before:
        r = s.result(log=False)
        print(f"  basis factorised on the device in {time.time() - t0:.1f} s (creation, upload included): "
              f"k = {r.dense_columns}, inverse {8 * r.dense_columns ** 2 / 1e9:.1f} GB", flush=True)
    status = "iter_limit"
    while status == "iter_limit" and time.time() - t0 < max_s:
        t1 = time.time()
        status = s.run(chunk)
        r = s.result(log=False)
        print(f"  {done_before + r.iterations:9d} pivots  {seconds_before + time.time() - t0:6.0f} s  {status:10s} k={r.dense_columns:6d}  "
              f"{chunk / (time.time() - t1):7.0f} it/s  objective {r.objective!r}  max_pivot_error "
              f"{r.max_pivot_error:.1e}  near ties {r.near_ties}  refactors {r.refactors}", flush=True)
    r = s.result(log=False)
print(f"sparse {m}x{ns}, {per_col} per column, seed {seed}: {status} after {done_before + r.iterations} pivots in "
      f"{seconds_before + time.time() - t0:.0f} s, objective {r.objective!r}, k = {r.dense_columns} "
      f"(inverse {8 * r.dense_columns ** 2 / 1e9:.1f} GB), refactorisations {r.refactors}, "
      f"max_pivot_error {r.max_pivot_error:.1e}")
if state_out and status == "iter_limit":
    os.makedirs(os.path.dirname(os.path.abspath(state_out)), exist_ok=True)
    np.savez(state_out, m=m, ns=ns, per_col=per_col, seed=seed, pivots=done_before + r.iterations,
             seconds=seconds_before + time.time() - t0, basis=r.basis, nonbasis=r.nonbasis, x=r.x,
             xbar=r.xbar, z=r.z, zbar=r.zbar)
    print(f"state written to {state_out}", flush=True)
a = sp.csc_matrix((val, ri, cp), shape=(m, ns))
xs, slack = np.zeros(ns), np.zeros(m)
for pos, var in enumerate(r.basis):
    if var < ns:
        xs[var] = r.x[pos]
    else:
        slack[var - ns] = r.x[pos]
y = np.zeros(m)
for pos, var in enumerate(r.nonbasis):
    if var >= ns:
        y[var - ns] = r.z[pos]
primal_res = float(np.abs(a @ xs + slack - b).max())
primal_neg = float(max(0.0, -xs.min(), -slack.min()))
dual_slack = a.T @ y - c
dual_neg = float(max(0.0, -dual_slack.min(), -y.min()))
pobj, dobj = float(c @ xs), float(b @ y)
print(f"certificate: |A x + s - b|max {primal_res:.2e}, most negative x/s {primal_neg:.2e}, most negative "
      f"dual slack / y {dual_neg:.2e}, c.x {pobj!r}, b.y {dobj!r}, gap (rel) "
      f"{abs(pobj - dobj) / max(1.0, abs(pobj)):.2e}, engine objective vs c.x (rel) "
      f"{abs(r.objective - pobj) / max(1.0, abs(pobj)):.2e}")

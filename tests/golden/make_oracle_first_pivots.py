"""First pivots of a seeded G1 LP under the CPU oracle (the C restatement of the reference's
arithmetic, oracle/) at sizes where one pivot costs the oracle half a minute to several minutes
(two dense LUs of B and of B^T from scratch per iteration, as src/simplex.rs:226-236 does):
tests/golden/oracle_first_pivots_<seed>_<m>x<ns>.json.

The oracle's state (basis, nonbasis, x, xbar, z, zbar) lives in the arrays handed to
ora_simplex_solve, so the solve is advanced ONE pivot per call and the fixture is rewritten after
every pivot: an interrupted run keeps what it has.  Each record is what
src/simplex.rs:274-306,308-330 decides -- (kind, entering, leaving) -- with mu* of that
iteration and the seconds of CPU the pivot took on one core of the machine that ran this.

  python3 tests/golden/make_oracle_first_pivots.py seed m ns pivots [--blocked] [--resume] [--dense-input]
                                                   [--sparse-per-col N]
  (BASELINE config 3: 1003 8192 16384 8  -- about 6.5 minutes per pivot)

--blocked: the twin library (oracle/dzg_oracle_blocked.c: Matrix::factorize applied block by
block on several cores, the same operations per element in the same order -- bit-equal factors,
tests/test_oracle_kats.py), some 20 s per pivot at 8192 rows; the file is then called
oracle_blocked_pivots_<seed>_<m>x<ns>.json and says so.  The literal run's pivots are the check
on it: both files exist for config 3 and tests/test_oracle_kats.py compares them.

--resume: continue from the checkpoint the last run left (the oracle's six state arrays after its
last pivot, gpurun_out/oracle_ckpt_<seed>_<m>x<ns>.npz -- scratch, not committed) instead of from
the first pivot.  A fixture on disk is only ever replaced by a LONGER one.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dantzig_amd import core  # noqa: E402  (host-side generator only, no GPU needed)
from oracle import oracle as ora  # noqa: E402

if __name__ == "__main__":
    seed, m, ns, pivots = (int(v) for v in sys.argv[1:5])
    blocked = "--blocked" in sys.argv
    the_lib = ora.blocked_lib() if blocked else ora.lib()
    path = os.path.join(ROOT, "tests", "golden",
                        f"oracle_{'blocked' if blocked else 'first'}_pivots_{seed}_{m}x{ns}.json")
    per_col = int(sys.argv[sys.argv.index("--sparse-per-col") + 1]) if "--sparse-per-col" in sys.argv else 0
    if per_col > 0:
        # generator G2 (BASELINE config 4): the reference's own CscMatrix over all columns, unit slack
        # columns appended; the oracle densifies the basis like the reference does (two m x m LUs)
        path = path.replace(".json", f"_csc{per_col}.json")
        cp, ri, val_s, b, c = core.gen_sparse_lp(seed, m, ns, per_col)
        n, q = ns + m, ns
        col_ptr = np.concatenate([ora._i64(cp), ora._i64(cp[-1] + 1 + np.arange(m))])
        row_idx = np.concatenate([ora._i64(ri), np.arange(m, dtype=np.int64)])
        val = np.concatenate([ora._f64(val_s), np.ones(m)])
        basis, nonbasis = np.arange(ns, ns + m, dtype=np.int64), np.arange(ns, dtype=np.int64)
        x, z = ora._f64(b).copy(), -ora._f64(c)
        xbar, zbar = np.ones(m), np.ones(q)
        cc = np.concatenate([ora._f64(c), np.zeros(m)])
        st = ora._Simplex(m, n, ora._p(col_ptr), ora._p(row_idx), ora._p(val), ora._p(cc), 0.0,
                          ora._p(basis), ora._p(nonbasis), ora._p(x), ora._p(xbar), ora._p(z), ora._p(zbar))
        a = None
    else:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    if per_col > 0:
        pass
    elif "--dense-input" in sys.argv:
        # the oracle's dense-structural input format (dzg_oracle.h, row_idx == NULL): the generator's
        # column-major block as it is, no 64-bit CSC beside it (34 GB at 32768 x 65536)
        n, q = ns + m, ns
        val = np.ascontiguousarray(a.T)  # (ns, m) C-contiguous == column-major m x ns: no copy
        assert val.base is a.base or val.base is a or np.shares_memory(val, a)
        basis, nonbasis = np.arange(ns, ns + m, dtype=np.int64), np.arange(ns, dtype=np.int64)
        x, z = ora._f64(b).copy(), -ora._f64(c)
        xbar, zbar = np.ones(m), np.ones(q)
        cc = np.concatenate([ora._f64(c), np.zeros(m)])
        col_ptr = np.zeros(1, dtype=np.int64)
        st = ora._Simplex(m, n, ora._p(col_ptr), None, ora._p(val), ora._p(cc), 0.0,
                          ora._p(basis), ora._p(nonbasis), ora._p(x), ora._p(xbar), ora._p(z), ora._p(zbar))
    else:
        sf = ora.stdform_from_dense(a, b, c)
        del a
        n, q = sf.n, sf.n - sf.m
        basis, nonbasis = ora._i64(sf.basis).copy(), ora._i64(sf.nonbasis).copy()
        x, z = ora._f64(sf.x).copy(), ora._f64(sf.z).copy()
        xbar, zbar = np.ones(m), np.ones(q)
        col_ptr, row_idx, val, cc = ora._i64(sf.col_ptr), ora._i64(sf.row_idx), ora._f64(sf.val), ora._f64(sf.c)
        st = ora._Simplex(m, n, ora._p(col_ptr), ora._p(row_idx), ora._p(val), ora._p(cc), float(sf.constant),
                          ora._p(basis), ora._p(nonbasis), ora._p(x), ora._p(xbar), ora._p(z), ora._p(zbar))
    ckpt = os.path.join(ROOT, "gpurun_out", f"oracle_ckpt_{'blocked' if blocked else 'first'}_{seed}_{m}x{ns}.npz")
    prior = None
    if "--resume" in sys.argv and os.path.exists(ckpt) and os.path.exists(path):
        ck = np.load(ckpt)
        with open(path) as f:
            prior = json.load(f)
        if int(ck["pivots"]) == len(prior["kind"]):
            basis[:], nonbasis[:], x[:], xbar[:], z[:], zbar[:] = (ck[k] for k in ("basis", "nonbasis", "x", "xbar", "z", "zbar"))
            print(f"resuming after pivot {len(prior['kind'])}", flush=True)
        else:
            prior = None
    existing = 0
    if os.path.exists(path):
        with open(path) as f:
            existing = len(json.load(f)["kind"])
    out = {"seed": seed, "m": m, "n_struct": ns,
           "generator": (f"G2 (dantzig_amd.core.gen_sparse_lp, {per_col} nonzeros per column)" if per_col > 0
                         else "G1 (dantzig_amd.core.gen_dense_lp)"),
           "source": ("oracle/dzg_oracle.c + dzg_oracle_blocked.c (libdzg_oracle_blocked.so)" if blocked
                      else "oracle/dzg_oracle.c") + ", one ora_simplex_solve(max_iter=1) call per pivot",
           "kind": [], "entering": [], "leaving": [], "mu": [], "seconds_per_pivot": []}
    if prior is not None:
        for key in ("kind", "entering", "leaving", "mu", "seconds_per_pivot"):
            out[key] = list(prior[key])
    log = (ora._Pivot * 1)()
    for k in range(len(out["kind"]), pivots):
        iters = C.c_int64(0)
        t0 = time.perf_counter()
        status = the_lib.ora_simplex_solve(C.byref(st), C.c_int64(1), C.byref(iters), log, C.c_int64(1))
        dt = time.perf_counter() - t0
        if iters.value != 1:
            out["status_after"] = ora.STATUS[status]
            break
        out["kind"].append(int(log[0].kind))
        out["entering"].append(int(log[0].entering))
        out["leaving"].append(int(log[0].leaving))
        out["mu"].append(float(log[0].mu))
        out["seconds_per_pivot"].append(round(dt, 1))
        if len(out["kind"]) > existing:  # (a fixture is only ever replaced by a longer one)
            with open(path + ".tmp", "w") as f:
                json.dump(out, f)
            os.replace(path + ".tmp", path)
            os.makedirs(os.path.dirname(ckpt), exist_ok=True)
            np.savez(ckpt + ".tmp.npz", pivots=len(out["kind"]), basis=basis, nonbasis=nonbasis, x=x, xbar=xbar,
                     z=z, zbar=zbar)
            os.replace(ckpt + ".tmp.npz", ckpt)
        print(f"pivot {k + 1}: {(log[0].kind, log[0].entering, log[0].leaving)} mu {log[0].mu!r} "
              f"in {dt:.1f} s", flush=True)

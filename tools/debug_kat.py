"""debug: run one solver KAT through Level 1 in FAST and STRICT, compare pivot logs with the oracle."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dantzig_amd import core
from oracle import oracle as ora


def corelp_from_stdform(sf):
    dense = ora.csc_to_dense(sf.m, sf.n, sf.col_ptr, sf.row_idx, sf.val)
    is_slack = np.zeros(sf.n, bool)
    is_slack[sf.basis] = True
    var_col = np.zeros(sf.n, np.int64)
    cols = []
    for v in range(sf.n):
        if is_slack[v]:
            r = int(np.nonzero(dense[:, v])[0][0])
            var_col[v] = -1 - r
        else:
            var_col[v] = len(cols)
            cols.append(dense[:, v])
    a = np.stack(cols, axis=1) if cols else np.zeros((sf.m, 0))
    return core.CoreLP(a=a, c=sf.c, basis=sf.basis, nonbasis=sf.nonbasis, x=sf.x, z=sf.z,
                       var_col=var_col, constant=sf.constant)


name = sys.argv[1] if len(sys.argv) > 1 else "nonneg_3"
kats = json.load(open(os.path.join(ROOT, "tests/golden/reference_kats.json")))
k = next(s for s in kats["solver"] if s["name"] == name)
sf = ora.build_standard_form(k["model"])
want = ora.simplex_solve(sf)
print("oracle", want.status, want.iterations, want.pivots)
lp = corelp_from_stdform(sf)
for num in (core.STRICT, core.FAST):
    got = core.solve(lp, numerics=num, max_iter=50)
    print("gpu", got.numerics, got.status, got.iterations, got.pivots)
    print("   x", got.x, "\n   xbar", got.xbar, "\n   z", got.z, "\n   zbar", got.zbar)
print("oracle x", want.x, "\n xbar", want.xbar, "\n z", want.z, "\n zbar", want.zbar)

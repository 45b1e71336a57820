"""Fill-in of a sparse LU on the k x k basis block A[R, S] of BASELINE config 4's family (uniformly
random pattern, 50 nonzeros per column over m = 50 000 rows, so 50 k / m per column inside the
block), measured with SuperLU (scipy.sparse.linalg.splu, COLAMD ordering, partial pivoting).
A unit diagonal is added so that the block is nonsingular like a basis block is.  CPU only.

  python3 tools/sparse_lu_fill.py [k values, comma separated]
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

m, per_col = 50_000, 50
ks = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1000,2000,4000,8000").split(",")]
rng = np.random.default_rng(1004)
print(f"family: {per_col} nonzeros per column over m = {m} rows; block k x k has {per_col}k/m per column")
for k in ks:
    inside = rng.binomial(per_col, k / m, size=k)          # entries of each column that fall in R
    rows = np.concatenate([rng.choice(k, n, replace=False) for n in inside] + [np.arange(k)])
    cols = np.concatenate([np.full(n, j) for j, n in enumerate(inside)] + [np.arange(k)])
    vals = np.concatenate([rng.uniform(-1, 1, inside.sum()), np.ones(k)])
    g = sp.csc_matrix((vals, (rows, cols)), shape=(k, k))
    g.sum_duplicates()
    t0 = time.time()
    lu = splu(g, permc_spec="COLAMD")
    dt = time.time() - t0
    fill = lu.L.nnz + lu.U.nnz
    print(f"k = {k:6d}: nnz(G) = {g.nnz:8d} ({g.nnz / k:5.1f} per column)   nnz(L+U) = {fill:10d} "
          f"= {fill / (k * k):6.1%} of k^2 ({fill * 12 / 1e6:8.1f} MB as CSC vs {8 * k * k / 1e6:8.1f} MB "
          f"for the dense k x k inverse)   SuperLU {dt:6.1f} s", flush=True)

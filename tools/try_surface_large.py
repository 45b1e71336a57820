"""Large models through the Python surface (Level 2 -> CSC path): a random sparse LP and a
degenerate transportation LP, checked against scipy/HiGHS."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dantzig_amd as dz
from scipy.optimize import linprog
import scipy.sparse as sp

def random_sparse(K=1500, M=600, per_row=6, seed=3):
    rng = np.random.default_rng(seed)
    xs = [dz.Variable(lb=0.0, ub=10.0) for _ in range(K)]
    c = rng.uniform(0.1, 1.0, K)
    rows, A = [], sp.lil_matrix((M, K))
    x0 = rng.uniform(0, 1, K)
    for r in range(M):
        idx = rng.choice(K, per_row, replace=False)
        coef = rng.uniform(0.1, 1.0, per_row)
        A[r, idx] = coef
        rows.append((idx, coef, float(coef @ x0[idx]) + rng.uniform(0.1, 1)))
    prob = dz.Maximize(sum(float(ci) * xi for ci, xi in zip(c, xs)))
    prob.subject_to([sum(float(cf) * xs[i] for i, cf in zip(idx, coef)) <= b for idx, coef, b in rows])
    ref = linprog(-c, A_ub=A.tocsr(), b_ub=[b for _, _, b in rows], bounds=(0, 10), method="highs")
    return prob, -ref.fun

def transportation(S=30, D=40, seed=4):
    rng = np.random.default_rng(seed)
    supply = rng.integers(20, 60, S).astype(float)
    demand = rng.integers(5, 25, D).astype(float)
    cost = rng.integers(1, 20, (S, D)).astype(float)
    x = [[dz.Variable.nonneg() for _ in range(D)] for _ in range(S)]
    prob = dz.Minimize(sum(cost[i][j] * x[i][j] for i in range(S) for j in range(D)))
    cons = [sum(x[i][j] for j in range(D)) <= supply[i] for i in range(S)]
    cons += [sum(x[i][j] for i in range(S)) >= demand[j] for j in range(D)]
    prob.subject_to(cons)
    A = np.zeros((S + D, S * D)); b = np.zeros(S + D)
    for i in range(S):
        A[i, i * D:(i + 1) * D] = 1; b[i] = supply[i]
    for j in range(D):
        A[S + j, j::D] = -1; b[S + j] = -demand[j]
    ref = linprog(cost.ravel(), A_ub=A, b_ub=b, bounds=(0, None), method="highs")
    return prob, ref.fun

for name, make in (("random_sparse", random_sparse), ("transportation", transportation)):
    t = time.time(); prob, want = make(); tb = time.time() - t
    t = time.time()
    try:
        sol = prob.solve()
        print(f"{name}: build {tb:.1f}s solve {time.time()-t:.2f}s objective {sol.objective_value!r} highs {want!r} "
              f"rel {abs(sol.objective_value-want)/max(1,abs(want)):.2e} iters {sol._solution.iterations} "
              f"numerics {sol._solution.numerics} shape {sol._solution.shape}", flush=True)
    except Exception as e:
        print(f"{name}: FAILED {type(e).__name__}: {e} (highs {want!r})", flush=True)

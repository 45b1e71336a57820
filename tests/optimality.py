"""Independent optimality certificate for  max c.x  st  A x + s = b,  x, s >= 0  (test helper).

Given only a basis, LAPACK (numpy) recomputes x_B = B^-1 b and y = B^-T c_B and reports primal
feasibility, dual feasibility and both objectives: by strong duality a basis that passes is
optimal whatever arithmetic found it.  Usable at sizes no CPU simplex reaches in test time.
"""
import numpy as np


def certificate(a, b, c, basis):
    """max c.x st Ax + s = b, x,s >= 0.  Returns dict of residuals for the basis `basis`."""
    m, ns = a.shape
    bm = np.zeros((m, m))
    cb = np.zeros(m)
    for pos, var in enumerate(basis):
        if var < ns:
            bm[:, pos] = a[:, var]
            cb[pos] = c[var]
        else:
            bm[var - ns, pos] = 1.0
    xb = np.linalg.solve(bm, b)
    y = np.linalg.solve(bm.T, cb)
    red = a.T @ y - c            # reduced costs of structurals (basic ones ~ 0)
    return {"primal_infeas": float(max(0.0, -xb.min())),
            "dual_infeas": float(max(0.0, -red.min(), -y.min())),
            "primal_obj": float(cb @ xb), "dual_obj": float(b @ y),
            "resid": float(np.abs(bm @ xb - b).max())}

"""CPU emulation of the column-sharded iteration protocol (dantzig_amd.h, dzg_shard_phase*),
one OS process per rank, records exchanged with torch.distributed (gloo).

Test infrastructure: the HIP kernels are replaced by the oracle's functions (lu_solve,
neg_t_dot, pivot rules), everything else -- column ownership (col_range), the record layout,
which rank proposes what in which phase, the deterministic merge (dzg_merge_candidates), the
replicated x side and the owner-maintained z side -- follows the product's design, so that the
multi-process logic can be checked against the single-process oracle without GPUs."""
import numpy as np

from dantzig_amd import core
from dantzig_amd.sharded import all_gather_records, col_range
from oracle import oracle as ora

EPS = 1e-12


def _exchange(rec, world):
    import torch

    send = torch.from_numpy(np.ascontiguousarray(rec))
    recv = torch.zeros(world * len(rec), dtype=torch.float64)
    all_gather_records(send, recv)
    return recv.numpy().reshape(world, len(rec))


def _merge(recs):
    w = core.merge_candidates([(r[0], int(r[1]), r[2], r[3], r[4]) for r in recs])
    return w


def solve_sharded(a, b, c, rank, world, max_iter=100000):
    """Returns (status, pivot log) as seen by this rank."""
    m, ns = a.shape
    n = ns + m
    begin, end = col_range(ns, rank, world)
    basis = np.arange(ns, n)
    nonbasis = np.arange(ns)
    x, xbar = b.astype(float).copy(), np.ones(m)
    z, zbar = -c.astype(float).copy(), np.ones(ns)
    full = np.concatenate([a, np.eye(m)], axis=1)
    cp, ri, val = ora.csc_from_dense(full)
    nrec = 8 + m
    log = []

    def owned(var):
        return var >= ns or begin <= var < end

    def column(var):
        return full[:, var]

    def record(ratio, pos, dzv=0.0, with_column=True):
        rec = np.zeros(nrec)
        rec[1] = -1
        if pos >= 0:
            var = nonbasis[pos]
            rec[:6] = [ratio, pos, z[pos], zbar[pos], dzv, var if var < ns else -1 - (var - ns)]
            if with_column and var < ns:
                rec[8:] = a[:, var]  # only the owner ever gets here for a structural variable
        return rec

    def entering_column(rec):
        code = int(rec[5])
        if code >= 0:
            return rec[8:].copy()
        e = np.zeros(m)
        e[-1 - code] = 1.0
        return e

    for _ in range(max_iter):
        # ---- phase 1: first-pivot proposal on the z side (owned columns + slack positions)
        mine = np.array([owned(v) for v in nonbasis])
        idx = np.nonzero(mine)[0]
        k = ora.find_first_pivot(z[idx], zbar[idx]) if len(idx) else -1
        pos = int(idx[k]) if k >= 0 else -1
        recs = _exchange(record(-z[pos] / zbar[pos] if pos >= 0 else 0.0, pos), world)
        # ---- phase 2: merge, status(), first half of the step
        w = _merge(recs)
        pj = int(recs[w][1]) if w >= 0 else -1
        pi = ora.find_first_pivot(x, xbar)
        if pj >= 0 and pi >= 0:
            primal, dual = -x[pi] / xbar[pi], recs[w][0]
            if primal <= EPS and dual <= EPS:
                return "optimal", log
            kind, mu = (ora.PRIMAL, dual) if primal < dual else (ora.DUAL, primal)
        elif pj >= 0:
            kind, mu = ora.PRIMAL, recs[w][0]
        elif pi >= 0:
            kind, mu = ora.DUAL, -x[pi] / xbar[pi]
        else:
            return "panic", log
        bm = full[:, basis]
        if kind == ora.PRIMAL:
            n_j = pj
            dx = ora.lu_solve(bm, entering_column(recs[w]))
            b_i = ora.find_second_pivot(mu, x, xbar, dx)
            if b_i < 0:
                return "unbounded", log
        else:
            b_i = pi
        e = np.zeros(m)
        e[b_i] = 1.0
        v = ora.lu_solve(ora.matrix_t(bm), e)
        dz = np.zeros(ns)
        dz[idx] = ora.neg_t_dot(cp, ri, val, nonbasis[idx], v)  # pricing of the owned part only
        if kind == ora.DUAL:
            k = ora.find_second_pivot(mu, z[idx], zbar[idx], dz[idx]) if len(idx) else -1
            p2 = int(idx[k]) if k >= 0 else -1
            ratio = dz[p2] / (z[p2] + mu * zbar[p2]) if p2 >= 0 else 0.0
            rec = record(ratio, p2, dz[p2] if p2 >= 0 else 0.0)
        else:
            rec = record(1.0, n_j, dz[n_j], with_column=False) if mine[n_j] else record(0.0, -1)
        recs2 = _exchange(rec, world)
        # ---- phase 3: merge, second half of the step, pivot
        if kind == ora.DUAL:
            w2 = _merge(recs2)
            if w2 < 0:
                return "infeasible", log
            n_j = int(recs2[w2][1])
            dx = ora.lu_solve(bm, entering_column(recs2[w2]))
        else:
            w2 = next(r for r in range(world) if int(recs2[r][1]) == n_j)
        zr, zbr, dzr = recs2[w2][2], recs2[w2][3], recs2[w2][4]
        t, tbar = x[b_i] / dx[b_i], xbar[b_i] / dx[b_i]
        s, sbar = zr / dzr, zbr / dzr
        log.append((kind, int(nonbasis[n_j]), int(basis[b_i])))
        keep = np.arange(m) != b_i
        x[keep] -= t * dx[keep]
        xbar[keep] -= tbar * dx[keep]
        x[b_i], xbar[b_i] = t, tbar
        upd = mine.copy()
        upd[n_j] = False
        z[upd] -= s * dz[upd]
        zbar[upd] -= sbar * dz[upd]
        z[n_j], zbar[n_j] = s, sbar  # every rank: the position changes owner with its variable
        basis[b_i], nonbasis[n_j] = nonbasis[n_j], basis[b_i]
    return "iter_limit", log

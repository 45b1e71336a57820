"""Column sharding.

CPU (gloo, world_size 2): the host-side exchange -- column partition, all-gather of the
candidate records through torch.distributed, deterministic merge (dzg_merge_candidates) --
checked against the oracle's global ratio test on the same data.
GPU: the complete sharded device path with all ranks inside one process on one GPU
(run_lockstep: the exchange is a device copy), checked against the oracle's pivot log."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as ora


def test_col_range_partitions():
    from dantzig_amd.sharded import col_range

    for ns in (1, 7, 16, 1000, 16384):
        for world in (1, 2, 3, 8):
            edges = [col_range(ns, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == ns
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1


def test_merge_candidates_rule():
    from dantzig_amd import core

    inf = float("inf")
    assert core.merge_candidates([(1.0, 5, 0, 0, 0), (2.0, 9, 0, 0, 0)]) == 1
    assert core.merge_candidates([(2.0, 9, 0, 0, 0), (2.0, 3, 0, 0, 0)]) == 1  # tie -> lowest pos
    assert core.merge_candidates([(0.0, -1, 0, 0, 0), (inf, 4, 0, 0, 0), (inf, 2, 0, 0, 0)]) == 2
    assert core.merge_candidates([(0.0, -1, 0, 0, 0)]) == -1
    assert core.merge_candidates([(float("nan"), 1, 0, 0, 0), (-3.0, 2, 0, 0, 0)]) == 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, seed, m, ns, out):
    import torch.distributed as dist

    from dantzig_amd import core
    from tests.shard_protocol import solve_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the generator is host code of the product library (no GPU needed)
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        status, log = solve_sharded(np.array(a), b, c, rank, world)
        out.put((rank, status, log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,m,ns", [(41, 12, 20), (42, 24, 48), (43, 32, 40)])
def test_sharded_protocol_over_gloo(seed, m, ns):
    """Two OS processes, one rank each, records over gloo: the sharded protocol reproduces the
    single-process oracle's pivot log on every rank."""
    import torch.multiprocessing as mp

    from dantzig_amd import core

    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, seed, m, ns, out))
             for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, status, log in results:
        assert status == want.status == "optimal"
        assert log == [(k, e, l) for k, e, l, _ in want.pivots]


@pytest.mark.gpu
@pytest.mark.parametrize("world,seed,m,ns", [(2, 31, 48, 96), (3, 32, 64, 100), (4, 33, 128, 256),
                                             (8, 34, 96, 200)])
def test_sharded_lockstep_matches_oracle(world, seed, m, ns):
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    solvers = make_lockstep(lp, world, poll_interval=16)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    log = [(k, e, l) for k, e, l, _ in want.pivots]
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == log
        assert res.basis.tolist() == want.basis.tolist()
        assert abs(res.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
    # every rank carries the same replicated x; z is only kept for owned / slack positions
    for res in results[1:]:
        assert np.array_equal(res.x, results[0].x)


@pytest.mark.gpu
def test_sharded_lockstep_from_column_blocks():
    """Each rank is given ONLY its own columns of A (gen_dense_lp_block + opts.a_is_block), as in
    a multi-GPU run of an LP too large to materialise per process: same pivots as the oracle."""
    from dantzig_amd import core
    from dantzig_amd.sharded import ShardedSolver, col_range, run_lockstep

    world, seed, m, ns = 3, 37, 72, 150
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    solvers = []
    try:
        for r in range(world):
            begin, end = col_range(ns, r, world)
            ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)
            lp = core.CoreLP.from_inequality_block(ab, bb, cb, begin, end)
            solvers.append(ShardedSolver(lp, r, world, poll_interval=16,
                                         stream=solvers[0].stream if solvers else 0))
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert abs(res.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))


@pytest.mark.gpu
def test_rccl_single_rank_loop():
    """The native RCCL loop with a communicator of one rank (all a 1-GPU box can host): loads
    librccl, ncclCommInitRank, two ncclAllGather per iteration on the solver's stream."""
    from dantzig_amd import core
    from dantzig_amd.sharded import ShardedSolver, comm_unique_id

    a, b, c = core.gen_dense_lp(seed=36, m=64, n_struct=128)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    with ShardedSolver(lp, 0, 1, poll_interval=16) as s:
        s.comm_init(comm_unique_id())
        status = s.run()
        res = s.result()
    assert status == want.status == "optimal"
    assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]


@pytest.mark.gpu
def test_sharded_budget_and_resume():
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = core.gen_dense_lp(seed=35, m=64, n_struct=128)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    solvers = make_lockstep(lp, 2, poll_interval=8)
    try:
        status, rounds = "iter_limit", 0
        while status == "iter_limit":
            status = run_lockstep(solvers, max_new_iters=50)
            rounds += 1
            assert rounds < 1000
        res = solvers[0].result()
    finally:
        for s in solvers:
            s.close()
    assert status == "optimal" and rounds > 1
    assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]


@pytest.mark.gpu
def test_sharded_sparse_lockstep_matches_oracle():
    """Column sharding of a CSC matrix: each rank holds only its block of columns (rebased
    column pointers); the exchange records carry the densified candidate column."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    m, ns, per_col = 80, 210, 4
    cp, ri, val, b, c = core.gen_sparse_lp(71, m, ns, per_col)
    col_ptr = np.concatenate([cp, cp[-1] + 1 + np.arange(m)])
    row_idx = np.concatenate([ri.astype(np.int64), np.arange(m)])
    sf = ora.StdForm(m=m, n=ns + m, col_ptr=col_ptr, row_idx=row_idx,
                     val=np.concatenate([val, np.ones(m)]), c=np.concatenate([c, np.zeros(m)]),
                     constant=0.0, basis=np.arange(ns, ns + m), nonbasis=np.arange(ns),
                     x=b.copy(), z=-c)
    want = ora.simplex_solve(sf)
    lp = core.CoreLP.from_csc(m, cp, ri, val, b, c)
    solvers = make_lockstep(lp, 3, poll_interval=16)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]


@pytest.mark.gpu
@pytest.mark.parametrize("world,replicate", [(4, False), (8, False), (8, True), (3, True)])
def test_sharded_whole_solve_follows_the_oracle_pivot_log(world, replicate):
    """BASELINE config 2 (1024 x 2048) column-sharded over 3 / 4 / 8 ranks in lockstep on one GPU:
    every rank takes the 21 642 pivots of the committed CPU-oracle log (tests/golden, 106 min of CPU)
    and ends in its basis -- eta flushes, compact-column appends and deletes, ownership changes of
    the pivot position and all.  Partitioned ranks run the separate phase kernels, ranks with a
    replicated matrix the fused ones (k_chain_pre / k_chain_post between the exchanges)."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_1002_1024x2048.npz"))
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=int(fx["m"]), n_struct=int(fx["n_struct"]))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    solvers = make_lockstep(lp, world, replicate=replicate, poll_interval=64)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == str(fx["status"]) == "optimal"
    for res in results:
        assert res.iterations == int(fx["iterations"])
        assert np.array_equal(np.array([p[0] for p in res.pivots]), fx["kind"])
        assert np.array_equal(np.array([p[1] for p in res.pivots]), fx["entering"])
        assert np.array_equal(np.array([p[2] for p in res.pivots]), fx["leaving"])
        assert np.array_equal(res.basis, fx["basis"])
        assert abs(res.objective - float(fx["objective"])) <= 1e-9 * abs(float(fx["objective"]))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 5])
def test_sharded_solve_is_bit_identical_to_the_unsharded_one(world):
    """The pricing kernel's sums do not depend on how columns are split over waves or ranks, the
    merges are deterministic and the basis side is replicated: a column-sharded FAST solve is the
    SAME computation as the single-GPU one -- pivot log, mu of every pivot, x and the objective
    bit for bit, not merely the same optimum."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = core.gen_dense_lp(seed=38, m=300, n_struct=700)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    single = core.solve(lp, numerics=core.FAST, poll_interval=16)
    solvers = make_lockstep(lp, world, poll_interval=16)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == single.status == "optimal"
    for res in results:
        assert res.pivots == single.pivots          # kind, entering, leaving AND mu, exactly
        assert np.array_equal(res.x, single.x) and np.array_equal(res.xbar, single.xbar)
        assert res.objective == single.objective
        assert res.near_ties == single.near_ties and res.min_margin == single.min_margin


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_replicated_matrix_sharding_matches_oracle_and_single_gpu(world):
    """opts.replicate_matrix: every rank holds all columns, only the pricing is split; the records
    are 64-byte headers.  Same pivots as the oracle, same bits as the single-GPU solve -- and as
    the partitioned sharding, whose records carry the columns."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = core.gen_dense_lp(seed=39, m=160, n_struct=420)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    single = core.solve(lp, numerics=core.FAST, poll_interval=16)
    solvers = make_lockstep(lp, world, replicate=True, poll_interval=16)
    try:
        assert all(s.record_doubles == 8 for s in solvers)
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert res.pivots == single.pivots and np.array_equal(res.x, single.x)
        assert res.objective == single.objective


@pytest.mark.gpu
def test_replicated_sharding_from_column_blocks_and_refactorisation():
    """Each rank is created with its own block only (a_is_block) and receives the other blocks
    through dzg_solver_upload_columns; a run before that is refused.  With every column resident a
    sharded solver can refactorise: a rebuild every 40 pivots leaves the pivot sequence alone."""
    from dantzig_amd import _ffi, core
    from dantzig_amd.sharded import ShardedSolver, col_range, run_lockstep

    world, seed, m, ns = 3, 40, 96, 250
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    solvers = []
    try:
        for r in range(world):
            begin, end = col_range(ns, r, world)
            ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)
            lp = core.CoreLP.from_inequality_block(ab, bb, cb, begin, end)
            solvers.append(ShardedSolver(lp, r, world, poll_interval=8, replicate=True,
                                         refactor_interval=40,
                                         stream=solvers[0].stream if solvers else 0))
        with pytest.raises(_ffi.DantzigAmdError, match="have not been uploaded"):
            run_lockstep(solvers)
        for r, s in enumerate(solvers):
            for other in range(world):
                if other != r:
                    ob, oe = col_range(ns, other, world)
                    s.upload_columns(ob, oe, np.asarray(a)[:, ob:oe])
        # every column exactly once: handing a block over twice is refused (it would leave another
        # one unset behind a column count that looks complete)
        ob, oe = col_range(ns, 1, world)
        with pytest.raises(_ffi.DantzigAmdError, match="overlap"):
            solvers[0].upload_columns(ob, oe, np.asarray(a)[:, ob:oe])
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert res.refactors >= 2
        assert abs(res.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))


# ------------------------------------------------------------------ several OS processes, one GPU
def _hip():
    """The HIP runtime the library itself uses, for the test's own device buffers."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    return hip


def _gpu_phase_worker(rank, world, port, seed, m, ns, out, shard_rows=False):
    """One rank = one OS process with its own solver on the one GPU.  The host drives
    dzg_shard_phase1 / 2 / 3 itself and moves the records with torch.distributed over gloo (device
    -> host -> all-gather -> device): the product's phase kernels in a real multi-process run."""
    import ctypes as C
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dantzig_amd import core
        from dantzig_amd.sharded import ShardedSolver, all_gather_records, col_range

        begin, end = col_range(ns, rank, world)
        ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)   # this rank's columns only
        lp = core.CoreLP.from_inequality_block(ab, bb, cb, begin, end)
        hip = _hip()
        with ShardedSolver(lp, rank, world, poll_interval=1, shard_rows=shard_rows) as s:
            nrec = s.record_doubles
            lo, hi = s.row_range()
            # (two receive buffers: phase 3 of a row-sharded rank still reads the records of exchange 1)
            send, recv, recv2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
            assert hip.hipMalloc(C.byref(send), 8 * nrec) == 0
            assert hip.hipMalloc(C.byref(recv), 8 * nrec * world) == 0
            assert hip.hipMalloc(C.byref(recv2), 8 * nrec * world) == 0
            h_send = torch.zeros(nrec, dtype=torch.float64)
            h_recv = torch.zeros(nrec * world, dtype=torch.float64)

            def exchange(to):
                assert hip.hipMemcpy(h_send.data_ptr(), send, 8 * nrec, 2) == 0      # device -> host
                all_gather_records(h_send, h_recv)
                assert hip.hipMemcpy(to, h_recv.data_ptr(), 8 * nrec * world, 1) == 0  # host -> device

            assert s.set_budget(0) == "running"
            status, pivots = "running", 0
            while status == "running" and pivots < 20000:
                s.phase1(send.value)
                exchange(recv)
                s.phase2(recv.value, send.value)
                exchange(recv2)
                s.phase3(recv2.value)
                status, pivots = s.poll()
            res = s.result()
            hip.hipFree(send)
            hip.hipFree(recv)
            hip.hipFree(recv2)
        # (a host that drives the phases itself gets no gather of x: each rank answers for its rows)
        out.put((rank, status, res.pivots, (lo, hi, res.x[lo:hi].tolist()), res.objective))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,seed,m,ns,shard_rows", [(2, 45, 96, 200, False), (3, 46, 64, 150, False),
                                                        (3, 47, 96, 230, True)])
def test_sharded_processes_on_one_gpu_exchange_over_gloo(world, seed, m, ns, shard_rows):
    """The column-sharded device path as SEVERAL OS PROCESSES (one rank each, all on this one GPU),
    the host driving dzg_shard_phase1 / 2 / 3 and exchanging the records with torch.distributed
    (gloo): every rank holds only its own column block (partitioned storage, a_is_block), the
    entering column travels in the records.  Every rank must take the oracle's pivots and hold the
    single-GPU FAST solve's numbers bit for bit.  (RCCL refuses two ranks on one device; the RCCL loop
    itself runs with one rank in test_rccl_single_rank_loop.)  shard_rows: the same with the basis side
    sharded by rows (csrc/k_rowshard.hip) -- the records carry a row of the inverse at their full
    width, each rank answers for its own rows of x."""
    import torch.multiprocessing as mp

    from dantzig_amd import core

    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    single = core.solve(core.CoreLP.from_inequality_form(a, b, c), numerics=core.FAST, poll_interval=1,
                        seven_launches=1)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_phase_worker, args=(r, world, port, seed, m, ns, out, shard_rows))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert want.status == single.status == "optimal"
    for _, status, pivots, x, objective in results:
        assert status == "optimal"
        assert [(k, e, l) for k, e, l, _ in pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert [tuple(p) for p in pivots] == [tuple(p) for p in single.pivots]   # mu too, exactly
        lo, hi, own = x
        assert own == single.x[lo:hi].tolist()
        if not shard_rows:
            assert (lo, hi) == (0, m) and objective == single.objective


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [False, True])
def test_partitioned_sharding_refactorises(sparse):
    """a PARTITIONED column-sharded solver (a rank holds its own column block
    only) refactorises too -- every rank gathers what it has of A[R, S] and A[L, S], the shares are
    summed over the ranks (one owner per column: exact; RCCL all-reduce in dzg_shard_run, a copy
    kernel in this lockstep harness), and every rank factorises the same matrix.  A rebuild every 40
    pivots: the oracle's pivots, and the bits of the single-GPU solve that refactorises at the same
    pivots (dense; the CSC single-GPU solver runs the sparse-basis path, another representation)."""
    from dantzig_amd import core
    from dantzig_amd.sharded import ShardedSolver, col_range, run_lockstep

    world, seed, m, ns = 3, 41, 96, 250
    if sparse:
        cp, ri, val, b, c = core.gen_sparse_lp(seed, m, ns, 6)
        import scipy.sparse as sp

        a = sp.csc_matrix((val, ri, cp), shape=(m, ns)).toarray()
        whole = core.CoreLP.from_csc(m, cp, ri, val, b, c)
    else:
        a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
        whole = core.CoreLP.from_inequality_form(a, b, c)
    want = ora.simplex_solve(ora.stdform_from_dense(np.asarray(a), b, c))
    single = None if sparse else core.solve(whole, numerics=core.FAST, poll_interval=8, refactor_interval=40)
    solvers = []
    try:
        for r in range(world):
            if sparse:
                lp = whole
            else:
                begin, end = col_range(ns, r, world)
                ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)
                lp = core.CoreLP.from_inequality_block(ab, bb, cb, begin, end)
            solvers.append(ShardedSolver(lp, r, world, poll_interval=8, refactor_interval=40,
                                         stream=solvers[0].stream if solvers else 0))
        assert all(s.record_doubles > 8 for s in solvers)   # partitioned: the column travels
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert res.refactors >= 2 and res.refactors == results[0].refactors
        assert abs(res.objective - want.objective) <= 1e-9 * max(1.0, abs(want.objective))
        if single is not None:
            assert res.refactors == single.refactors
            assert res.pivots == single.pivots and np.array_equal(res.x, single.x)
            assert res.max_pivot_error == single.max_pivot_error


@pytest.mark.gpu
@pytest.mark.parametrize("replicate,shard_rows", [(False, False), (True, False), (False, True), (True, True)])
def test_sharded_warm_start_from_a_non_slack_basis(replicate, shard_rows):
    """A column-sharded solver created on a basis that is not the slack basis (the state a result
    handed back: dzg_lp.xbar / zbar) factorises it at its first run, the ranks exchanging their
    basic columns (partitioned) or after the last upload (replicated, a_is_block): the rest of the
    solve is, bit for bit, the single-GPU solver's continuation from the same state -- with the basis
    side replicated and with it sharded by rows (each rank created from its own column block, as
    `bench.py --gpus N` creates them)."""
    from dantzig_amd import core
    from dantzig_amd.sharded import ShardedSolver, col_range, run_lockstep

    world, seed, m, ns = 4, 42, 128, 300
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    whole = core.CoreLP.from_inequality_form(a, b, c)
    with core.Solver(whole, numerics=core.FAST, poll_interval=8) as s:
        assert s.run(150) == "iter_limit"
        mid = s.result()
    assert mid.dense_columns > 20
    cont = core.solve(core.resumed_from(whole, mid), numerics=core.FAST, poll_interval=8)
    assert cont.status == "optimal" and cont.refactors == 1
    solvers = []
    try:
        for r in range(world):
            begin, end = col_range(ns, r, world)
            ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)
            lp = core.resumed_from(core.CoreLP.from_inequality_block(ab, bb, cb, begin, end), mid)
            solvers.append(ShardedSolver(lp, r, world, poll_interval=8, replicate=replicate,
                                         shard_rows=shard_rows,
                                         stream=solvers[0].stream if solvers else 0))
        if replicate:
            for r, s in enumerate(solvers):
                for other in range(world):
                    if other != r:
                        ob, oe = col_range(ns, other, world)
                        s.upload_columns(ob, oe, np.asarray(a)[:, ob:oe])
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == "optimal"
    for res in results:
        assert res.refactors == 1
        assert res.pivots == cont.pivots and np.array_equal(res.x, cont.x)
        assert res.objective == cont.objective and np.array_equal(res.basis, cont.basis)


# ------------------------------------------------------------------ row-sharded basis side
@pytest.mark.gpu
@pytest.mark.parametrize("world,replicate", [(2, False), (3, True), (4, False), (8, True), (8, False)])
def test_row_sharded_solve_is_bit_identical_to_the_unsharded_one(world, replicate):
    """opts.shard_rows (csrc/k_rowshard.hip): rank r owns a block of ROWS of x, xbar, dx, the compact
    inverse and the eta columns as well as its block of columns; FTRAN, the x-side ratio test, the
    flush and the update touch those rows only, and an x-side candidate travels with its row of the
    inverse.  Same two exchanges, the same computation: the oracle's pivot log, and the single-GPU
    solve's mu of every pivot, x, xbar, basis, objective and monitor bit for bit -- partitioned and
    replicated matrix storage, row slices that leave some ranks without rows (8 ranks, 160 rows:
    slices of 32)."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    a, b, c = core.gen_dense_lp(seed=43, m=160, n_struct=420)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    single = core.solve(lp, numerics=core.FAST, poll_interval=16)
    solvers = make_lockstep(lp, world, replicate=replicate, shard_rows=True, poll_interval=16)
    try:
        status = run_lockstep(solvers)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == want.status == "optimal"
    for res in results:
        assert [(k, e, l) for k, e, l, _ in res.pivots] == [(k, e, l) for k, e, l, _ in want.pivots]
        assert res.pivots == single.pivots                      # mu of every pivot included
        assert np.array_equal(res.x, single.x) and np.array_equal(res.xbar, single.xbar)
        assert np.array_equal(res.basis, single.basis) and res.objective == single.objective
        assert res.max_pivot_error == single.max_pivot_error and res.min_margin == single.min_margin
        assert res.near_ties == single.near_ties


@pytest.mark.gpu
def test_row_sharded_on_integer_lps_and_terminal_verdicts():
    """Ties, zero pivots, unbounded and infeasible verdicts, near-tie bookkeeping: 40 small-integer
    and 0/1 LPs (and continuous ones) row-sharded over 3 ranks against the single-GPU FAST solve --
    status, pivot log with mu, near-tie record, bit for bit."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep
    from tests.lp_families import make_lp

    seen = set()
    for seed in range(9500, 9540):
        a, b, c = make_lp(seed, seed % 3, 3, 50)
        lp = core.CoreLP.from_inequality_form(a, b, c)
        single = core.solve(lp, numerics=core.FAST, poll_interval=8, max_iter=3000)
        solvers = make_lockstep(lp, 3, shard_rows=True, replicate=bool(seed & 1), poll_interval=8,
                                max_iter=3000)
        try:
            status = run_lockstep(solvers)
            results = [s.result() for s in solvers]
        finally:
            for s in solvers:
                s.close()
        seen.add(status)
        for res in results:
            assert (res.status, res.pivots) == (single.status, single.pivots), seed
            assert (res.near_ties, res.first_near_tie, res.min_margin) == \
                (single.near_ties, single.first_near_tie, single.min_margin), seed
            assert np.array_equal(res.x, single.x), seed
    assert {"optimal", "unbounded", "infeasible"} <= seen


@pytest.mark.gpu
@pytest.mark.parametrize("replicate", [False, True])
def test_row_sharded_whole_solve_with_refactorisation_and_resume(replicate):
    """BASELINE config 2 (1024 x 2048), 4 row-sharded ranks: a refactorisation every 3 000 pivots
    (partitioned: the ranks exchange their basic columns), the solve cut into budgeted runs (7 000 pivots each: x and
    xbar are gathered at every return and the rows' owners carry on from their own copies), the
    row-wise pricing pass and -- beyond k = rows_T -- the column pass, both re-derived for the
    entering column by every rank in a primal step: the oracle's 21 642 pivots, and the single-GPU
    solve with the same refactorisations bit for bit."""
    from dantzig_amd import core
    from dantzig_amd.sharded import make_lockstep, run_lockstep

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_pivots_1002_1024x2048.npz"))
    a, b, c = core.gen_dense_lp(seed=int(fx["seed"]), m=int(fx["m"]), n_struct=int(fx["n_struct"]))
    lp = core.CoreLP.from_inequality_form(a, b, c)
    single = core.solve(lp, numerics=core.FAST, poll_interval=50, refactor_interval=3000)
    assert single.price_pass_used == 3          # (k crosses the rule's threshold in this solve)
    solvers = make_lockstep(lp, 4, replicate=replicate, shard_rows=True, poll_interval=50,
                            refactor_interval=3000)
    try:
        status, runs = "iter_limit", 0
        while status == "iter_limit":
            status = run_lockstep(solvers, 7000)
            runs += 1
            mid = [s.result(log=False) for s in solvers]
            assert all(np.array_equal(r.x, mid[0].x) and np.array_equal(r.xbar, mid[0].xbar) for r in mid)
        results = [s.result() for s in solvers]
    finally:
        for s in solvers:
            s.close()
    assert status == "optimal" and runs == 4
    for res in results:
        assert res.iterations == int(fx["iterations"])
        assert np.array_equal(np.array([p[1] for p in res.pivots]), fx["entering"])
        assert np.array_equal(np.array([p[2] for p in res.pivots]), fx["leaving"])
        assert res.refactors == single.refactors >= 6
        assert res.pivots == single.pivots and np.array_equal(res.x, single.x)
        assert res.objective == single.objective and res.max_pivot_error == single.max_pivot_error

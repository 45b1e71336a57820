"""Column-sharded solve over several GPUs, one process per GPU.

The iteration loop is native (csrc/engine.hip, dzg_shard_run):

    phase1 -> ncclAllGather -> phase2 -> ncclAllGather -> phase3        (one simplex iteration)

on one HIP stream per rank, RCCL over xGMI between the GPUs of a node.  Every rank holds the
replicated basis inverse and x-side vectors and runs the same O(m k) basis kernels on identical
inputs; the matrix, z and the pricing pass are split by column ownership.  All ranks merge the
same exchange records with the same deterministic rule (largest ratio, lowest global position:
the reference's first-wins scan, src/simplex.rs:432-435,:456-459), so they take identical
decisions without a broadcast.

Python only bootstraps: torch.distributed (gloo, CPU) hands the 128-byte ncclUniqueId from
rank 0 to the other ranks and provides the barriers around the timed region.  torch.cuda is
not used: the process talks to the GPU through one HIP runtime only (the library's).
"""
from __future__ import annotations

import ctypes as C
import json
import time

import numpy as np

from . import _ffi, core


def col_range(n_struct: int, rank: int, world: int) -> tuple[int, int]:
    """Balanced contiguous column block of `rank`."""
    base, rem = divmod(n_struct, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class ShardedSolver(core.Solver):
    """One rank's share of a column-sharded solve."""

    def __init__(self, lp: core.CoreLP, rank: int, world: int, stream: int = 0,
                 replicate: bool = False, shard_rows: bool = False, **opts):
        """replicate=True (dense matrices): every rank keeps all structural columns in its HBM and
        only the pricing is split, so the exchange records shrink to their headers.  A block LP
        (CoreLP.from_inequality_block) then takes the other ranks' columns through upload_columns()
        before the first run.  shard_rows=True (dense matrices): the basis side is sharded too -- a
        rank owns a block of rows of x, the compact inverse and the eta file, FTRAN / the flush /
        the update touch those rows only and the x-side candidates travel with their row of the
        inverse (csrc/k_rowshard.hip); x and xbar are complete on every rank when a run returns."""
        begin, end = col_range(lp.n_struct, rank, world)
        self.rank, self.world = rank, world
        self.col_begin, self.col_end = begin, end
        super().__init__(lp, numerics=core.FAST, rank=rank, world=world, col_begin=begin,
                         col_end=end, stream=stream or None,
                         replicate_matrix=1 if (replicate and lp.a is not None) else 0,
                         shard_rows=1 if (shard_rows and lp.a is not None and world > 1) else 0, **opts)
        self.record_doubles = int(_ffi.lib().dzg_shard_record_doubles(self._h))

    def row_range(self) -> tuple[int, int]:
        """The rows of x / xbar this rank keeps while a row-sharded solve runs (shard_rows: slices of
        ceil16(ceil(m / world)) rows, include/dantzig_amd.h); everything otherwise.  dzg_shard_run and
        the lockstep loop complete x on every rank when they return; a host that drives the phases
        itself reads each rank's own rows."""
        m = self._lp.m
        if not getattr(self._opts, "shard_rows", 0):
            return 0, m
        per = max(16, (-(-m // self.world) + 15) // 16 * 16)
        lo = min(m, self.rank * per)
        return lo, min(m, lo + per)

    def upload_columns(self, begin: int, end: int, a_block) -> None:
        """Columns [begin, end) of A (an (m, end - begin) array) of a replicated solver."""
        a_cm = np.ascontiguousarray(np.asarray(a_block, dtype=np.float64).T)
        rc = _ffi.lib().dzg_solver_upload_columns(self._h, int(begin), int(end), _ffi.ptr(a_cm),
                                                  int(a_cm.shape[1]) if a_cm.ndim == 2 else 1)
        _ffi.check(rc, "dzg_solver_upload_columns")

    @property
    def stream(self) -> int:
        return int(_ffi.lib().dzg_solver_stream(self._h) or 0)

    def comm_init(self, unique_id: bytes) -> None:
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _ffi.check(_ffi.lib().dzg_shard_comm_init(self._h, buf), "dzg_shard_comm_init")

    def run(self, max_new_iters: int = 0) -> str:
        rc = _ffi.lib().dzg_shard_run(self._h, int(max_new_iters))
        _ffi.check(rc, "dzg_shard_run")
        return core.STATUS_NAMES[rc]

    # -- a host that moves the exchange records itself (any collective layer) drives the three
    # enqueue-only phases of an iteration; send / recv are DEVICE addresses it owns
    def set_budget(self, max_new_iters: int = 0) -> str:
        rc = _ffi.lib().dzg_solver_set_budget(self._h, int(max_new_iters))
        _ffi.check(rc, "dzg_solver_set_budget")
        return core.STATUS_NAMES.get(rc, str(rc))

    def phase1(self, send_dev: int) -> None:
        _ffi.check(_ffi.lib().dzg_shard_phase1(self._h, C.c_void_p(send_dev)), "dzg_shard_phase1")

    def phase2(self, recv_dev: int, send_dev: int) -> None:
        _ffi.check(_ffi.lib().dzg_shard_phase2(self._h, C.c_void_p(recv_dev), C.c_void_p(send_dev)),
                   "dzg_shard_phase2")

    def phase3(self, recv_dev: int) -> None:
        _ffi.check(_ffi.lib().dzg_shard_phase3(self._h, C.c_void_p(recv_dev)), "dzg_shard_phase3")

    def comm_size(self) -> int:
        rc = _ffi.lib().dzg_shard_comm_size(self._h)
        _ffi.check(rc, "dzg_shard_comm_size")
        return int(rc)

    def poll(self) -> tuple[str, int]:
        st, it = C.c_int32(0), C.c_int64(0)
        _ffi.check(_ffi.lib().dzg_solver_poll(self._h, C.byref(st), C.byref(it)), "poll")
        return core.STATUS_NAMES.get(st.value, str(st.value)), int(it.value)


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _ffi.check(_ffi.lib().dzg_comm_unique_id(buf), "dzg_comm_unique_id")
    return buf.raw


def make_lockstep(lp: core.CoreLP, world: int, replicate: bool = False, shard_rows: bool = False,
                  **opts) -> list:
    """All ranks of a sharded solve inside ONE process on ONE GPU, sharing one stream."""
    first = ShardedSolver(lp, 0, world, replicate=replicate, shard_rows=shard_rows, **opts)
    return [first] + [ShardedSolver(lp, r, world, stream=first.stream, replicate=replicate,
                                    shard_rows=shard_rows, **opts)
                      for r in range(1, world)]


def run_lockstep(solvers: list, max_new_iters: int = 0) -> str:
    """Native lockstep loop (dzg_shard_run_lockstep): the exchange is a device copy.  This is
    how the sharded device path is tested on a single-GPU box."""
    arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    rc = _ffi.lib().dzg_shard_run_lockstep(arr, len(solvers), int(max_new_iters))
    _ffi.check(rc, "dzg_shard_run_lockstep")
    return core.STATUS_NAMES[rc]


# ------------------------------------------------------------------ bench.py, N > 1
def _measure_sharded(dist, torch, rows, cols, seed, price_name, steps, warmup, rank, world,
                     local_rank, replicate: bool, shard_rows: bool = True, warm_k: int = 0) -> dict:
    """One dense G1 workload, column-sharded over `world` ranks.  Every rank generates only its own
    column block (dzg_gen_dense_lp_block: bit-identical to that slice of the whole LP, b and c
    complete), so no process ever holds the whole matrix on the host.  Collective: every rank
    returns.

    replicate=False -- the PARTITIONED storage BASELINE.json's north star names: a rank's HBM holds
    its column block only, the entering column travels in the exchange records.
    replicate=True -- every rank also receives the other ranks' blocks (one at a time) and keeps
    the whole matrix; only the pricing is split.
    shard_rows -- the basis side is sharded by rows too (csrc/k_rowshard.hip): FTRAN, the flush and
    the update divide by the rank count like the pricing pass; False: the round-3 form, the basis
    side replicated on every rank.
    warm_k > 0 -- the solve starts from the basis "first warm_k structural columns" (core.warm_started:
    the deep regime of a solve, reached without the pivots that lead there); the ranks factorise it
    together at the first run (partitioned: they exchange their basic columns)."""
    begin, end = col_range(cols, rank, world)
    t_gen = time.perf_counter()
    a, b, c = core.gen_dense_lp_block(seed, rows, cols, begin, end)
    lp = core.CoreLP.from_inequality_block(a, b, c, begin, end)
    if warm_k > 0:
        lp = core.warm_started(lp, warm_k)
    t_gen = time.perf_counter() - t_gen
    price = {"auto": core.PRICE_AUTO, "seq": core.PRICE_SEQ, "wave": core.PRICE_WAVE,
             "tree": core.PRICE_TREE}[price_name]
    # poll interval 50 divides the default warm-up and step counts: no partial batches
    solver = ShardedSolver(lp, rank, world, device=local_rank, price_kernel=price,
                           profile=(1 << _ffi.K_PRICE) | (EVENT_STRIDE << 16), poll_interval=50, replicate=replicate,
                           shard_rows=shard_rows)
    try:
        del a, lp
        # replicated: the other ranks' blocks are generated and uploaded one at a time (host peak:
        # two blocks)
        for other in range(world if replicate else 0):
            if other != rank:
                ob, oe = col_range(cols, other, world)
                blk, _, _ = core.gen_dense_lp_block(seed, rows, cols, ob, oe)
                solver.upload_columns(ob, oe, blk)
                del blk
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(uid, src=0)
        solver.comm_init(uid.numpy().tobytes())
        nranks = solver.comm_size()

        status = "iter_limit"
        t_warm = time.perf_counter()
        if warmup > 0:  # (a warm start's factorisation happens here, outside the timed region)
            status = solver.run(warmup)
        it0 = solver.poll()[1]          # poll synchronises the stream
        t_warm = time.perf_counter() - t_warm
        r_warm = solver.result(log=False)
        dist.barrier()
        t0 = time.perf_counter()
        if status == "iter_limit":
            status = solver.run(steps)
        it1 = solver.poll()[1]
        dist.barrier()
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        res = solver.result(log=False)
        pb = torch.tensor([res.price_bytes], dtype=torch.float64)
        dist.all_reduce(pb, op=dist.ReduceOp.SUM)
        record_bytes = 8 * solver.record_doubles
        # per-phase split, OUTSIDE the timed region (event stamps cost time): 200 more pivots with
        # every phase and both all-gathers bracketed by HIP events on this rank's stream
        phases = None
        if status == "iter_limit":
            solver.set_profile((1 << _ffi.K_COUNT) - 1)
            if solver.run(200) in ("iter_limit", "optimal"):
                rp = solver.result(log=False)
                n = max(rp.iterations - res.iterations, 1)
                if shard_rows:
                    labels = {"status": "propose record 1 (z and x first pivots, column, row of the inverse)",
                              "exchange1": "all-gather 1",
                              "ftran": "merge + status + (primal) FTRAN on the own rows / (dual) BTRAN row",
                              "price": "(dual) pricing of the own columns",
                              "ratio": "propose record 2", "exchange2": "all-gather 2",
                              "update": "merge + (primal) pricing / (dual) FTRAN on the own rows + books "
                                        "+ update + flush"}
                else:
                    labels = {"status": "propose first-pivot record", "exchange1": "all-gather 1",
                              "ftran": "merge + status + FTRAN + BTRAN", "price": "pricing (own columns)",
                              "ratio": "propose second record", "exchange2": "all-gather 2",
                              "update": "merge + (dual) FTRAN + pivot + update + flush"}
                mine = torch.tensor([1e3 * (rp.kernel_ms[k] - res.kernel_ms[k]) / n for k in labels],
                                    dtype=torch.float64)
                worst = mine.clone()
                dist.all_reduce(worst, op=dist.ReduceOp.MAX)
                phases = {"pivots": n, "us_per_iteration_rank0": dict(zip(labels.values(),
                                                                         [round(x, 2) for x in mine.tolist()])),
                          "us_per_iteration_max_over_ranks": dict(zip(labels.values(),
                                                                      [round(x, 2) for x in worst.tolist()])),
                          "note": "an all-gather's time runs from the moment this rank's stream "
                                  "reaches it: waiting for a slower rank is counted there"}
    finally:
        solver.close()
    dt = float(elapsed.item())
    done = it1 - it0
    # algorithmic bytes per pricing pass (one per pivot) over the HIP-event time of the timed passes
    # (every EVENT_STRIDE-th iteration of a batch: an event pair costs idle GPU time, bench.py)
    d_bytes = res.price_bytes - r_warm.price_bytes
    d_ms = res.kernel_ms["price"] - r_warm.kernel_ms["price"]
    d_launch = res.kernel_launches["price"] - r_warm.kernel_launches["price"]
    achieved = (d_bytes / max(done, 1) / 1e9) / max(d_ms / max(d_launch, 1) / 1e3, 1e-12) if d_launch > 0 else float("nan")
    return {
        "metric": "simplex_iterations_per_sec",
        "value": done / dt if dt > 0 else float("nan"),
        "unit": "iterations/s", "n_gpus": world, "steps": done, "warmup": it0,
        "ms_per_step": 1e3 * dt / max(done, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"dense random LP {rows}x{cols} fp64, generator G1 seed {seed}, "
                        f"column-sharded over {world} GPUs"
                        + (f", warm-started from a basis of {warm_k} structural columns" if warm_k else ""),
            "numerics": "fast", "price_kernel": price_name,
            "status_after_timed_region": status, "requested_steps": steps,
            "k_at_start": r_warm.dense_columns, "k_at_end": res.dense_columns,
            "exchanges_per_iteration": 2, "record_bytes_max": record_bytes,
            "matrix": ("replicated on every rank (pricing split by column block)"
                       if replicate else
                       "partitioned by column block (a rank holds its block only; the entering "
                       "column travels in the exchange records)"),
            "basis_side": ("sharded by rows (a rank owns m / N rows of x, the compact inverse and the eta "
                           "file; x-side candidates travel with their row of the inverse)" if shard_rows
                           else "replicated on every rank"),
            "collective": "ncclAllGather (RCCL) of one record per rank",
            "nranks_ncclCommCount": nranks,
            "lp_generation_s": round(t_gen, 3),
            "warmup_s_including_any_factorisation": round(t_warm, 3),
            "refactors": res.refactors,
            "max_pivot_error": res.max_pivot_error,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "pricing pass of rank 0 over its column block: row-wise (k_price_rows + "
                      "k_price_rows_finish) while k < 0.93 m n_s / (n_s + m), k_price_tree beyond; "
                      "k = %d at the end of the timed region%s" % (
                          res.dense_columns, " (dual steps only: a primal step of a row-sharded rank "
                                             "prices inside its last phase)" if shard_rows else ""),
            "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": None, "avg_launch_us": 1e3 * d_ms / max(d_launch, 1), "launches_timed": d_launch,
        },
        "cpu_baseline": None,
        "phases": phases,
        "pricing_bytes_all_ranks": float(pb.item()),
    }


EVENT_STRIDE = 8  # pricing passes per timed one (opts.profile bits 16..23), as in bench.py


DEEP = {"rows": 32768, "cols": 65536, "seed": 1005, "warm_k": 16384, "steps": 300, "warmup": 50}


def bench_main(args, rank: int, world: int, local_rank: int) -> int:
    import datetime

    import os
    import sys

    import torch
    import torch.distributed as dist

    # stdout carries exactly one JSON line: whatever gloo / RCCL print from native code while
    # they start up (version banners) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    _ffi.require_gpu()
    # gloo: bootstrap (ncclUniqueId) + barriers only; a rank that dies must not hang the others
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=900))
    # `value`: the PARTITIONED storage mode, the one BASELINE.json's north star states ("constraint
    # matrix column-block partitioned across the 8 GPUs"), the basis side sharded by rows (round 4);
    # "replicated": the same workload with the whole matrix in every rank's HBM (it fits: 1 GB / 17
    # GB of 288); "basis_replicated": the round-3 form (every rank streams the whole inverse).
    sub = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "phases")

    def run(rows, cols, seed, steps, warmup, **kw):
        return _measure_sharded(dist, torch, rows, cols, seed, args.price, steps, warmup, rank, world,
                                local_rank, **kw)

    out = run(args.rows, args.cols, args.seed, args.steps, args.warmup, replicate=False)
    rep = run(args.rows, args.cols, args.seed, args.steps, args.warmup, replicate=True)
    out["replicated"] = {k: rep[k] for k in sub}
    old = run(args.rows, args.cols, args.seed, args.steps, args.warmup, replicate=False, shard_rows=False)
    out["basis_replicated"] = {k: old[k] for k in sub}
    profiled = ("ROCP_TOOL_LIBRARIES" in os.environ
                or "rocprofiler" in os.environ.get("LD_PRELOAD", ""))
    default_workload = (args.rows == 8192 and args.cols == 16384
                        and not getattr(args, "no_secondary", False) and not profiled)
    if default_workload:
        # config 5: the LP the north star's 8-GPU target is quoted on (bench.py reports the same
        # workload on one GPU under the same key), from the slack basis ...
        sec = run(32768, 65536, 1005, 300, 50, replicate=False)
        out["secondary"] = {k: sec[k] for k in sub}
        # ... and DEEP in its solve, where sharding is meant to pay: warm-started from a basis of
        # 16 384 structural columns (the compact inverse is 4.3 GB, a pricing pass 8.6 GB: bench.py's
        # one-GPU line carries the same block), against the round-3 form on the same state.
        # (Partitioned storage only at this size: a replicated run has every rank generate all 17 GB.)
        dp = DEEP
        deep = run(dp["rows"], dp["cols"], dp["seed"], dp["steps"], dp["warmup"], replicate=False,
                   warm_k=dp["warm_k"])
        out["deep"] = {k: deep[k] for k in sub}
        deep = run(dp["rows"], dp["cols"], dp["seed"], dp["steps"], dp["warmup"], replicate=False,
                   shard_rows=False, warm_k=dp["warm_k"])
        out["deep"]["basis_replicated"] = {k: deep[k] for k in sub}
    if rank == 0 and world == 1 and not getattr(args, "no_cpu_baseline", False):
        # the reference's algorithm on this box's host (bench.py: cpu_baseline) -- at N = 1 only: with
        # more ranks the others would sit at the barrier below for it
        import bench

        at_size = (args.cols, args.seed, args.cpu_size_pivots) if args.cpu_size_pivots > 0 else None
        out["cpu_baseline"] = bench.cpu_baseline(args.cpu_sample_rows, 2 * args.cpu_sample_rows, 1002,
                                                 args.cpu_sample_pivots, args.rows, args.cpu_anchor_rows, at_size)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    return 0


def all_gather_records(send, recv) -> None:
    """recv[r] <- rank r's record, for every rank, with torch.distributed (any backend).
    For hosts that drive dzg_shard_phase1/2/3 themselves instead of dzg_shard_run."""
    import torch.distributed as dist

    dist.all_gather_into_tensor(recv, send)

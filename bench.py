#!/usr/bin/env python3
"""bench.py -- simplex iterations/s and pricing-kernel HBM GB/s on MI355X.

Metric (BASELINE.json): simplex iterations/sec + achieved HBM GB/s on the 8192 x 16384 dense
fp64 LP (generator G1, seed 1003).  One "step" = one executed pivot
(status + primal/dual step).  The LP is resident in HBM before the timed region starts
(dzg_solver_create uploads it); W warm-up pivots run first, then exactly K pivots are timed
between barriers, continuing the same solve trajectory.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rows M --cols NS --seed S]
                  [--price tree|seq|wave] [--no-cpu-baseline] [--no-pmc-traffic] [--no-secondary]
                  [--no-late] [--late-pivots N] [--warm-k K] [--cpu-size-pivots N]

`value` is the rate of the K pivots that follow the warm-up (an almost empty basis inverse); the
"late", "deep" and "end" blocks time K more pivots of the same solve after 20 000, 150 000 and
400 000 pivots, with the basis inverse grown (config.k_at_* / <block>.k_at_* = its dense columns)
and a per-kernel-class split; "whole_solve" is the solve to optimality.  The pricing pass is timed
with HIP events on the solver's stream around every 8th pass (roofline.launches_timed) and its HBM
traffic counted by rocprofv3 --pmc children (the timed region's pivots; and, warm-started, the deep
and end blocks' widths).  "cpu_baseline": the CPU oracle on this host -- real pivots of the same LP
at the same size by its blocked twin on 16 threads, the literal one-core loop beside it.

The default invocation also measures config 5 (32768 x 65536, the LP the multi-GPU target is quoted
on) and reports it under "secondary"; `value` is always the 8192 x 16384 LP.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# (before anything loads an OpenMP runtime: the CPU baseline's at-size leg runs the oracle's blocked
# twin on this many threads -- a GPU box gives one GPU's job 16 host cores)
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F64_PEAK_TFLOPS = 46.9  # v_mfma_f64_16x16x4_f64, measured: profiles/r02_mfma_f64_peak_microbench.txt
MFMA_F64_DATASHEET_TFLOPS = 78.6  # AMD's datasheet figure for fp64 matrix (SURVEY 8(d) names it)


# The reference's algorithm (a dense LU of B and of B^T from scratch in every iteration) is far
# from its flop model at these sizes: the 512-MB working set of one LU at 8192 rows leaves every
# cache, and the row-major elimination streams it once per step.  So the CPU figure at the
# benchmark size is not modelled: the oracle is timed LIVE on this host at two sizes -- 100 pivots
# at 1024 rows and one real pivot at 4096 rows -- and the exponent measured between them carries
# the rate to the benchmark size.  Real pivots at 8192 rows, timed once in the build container on
# one core, are on record beside it (tests/golden/oracle_first_pivots_1003_8192x16384.json:
# seconds_per_pivot, the fixture the GPU tests follow).
CPU_RECORD = "tests/golden/oracle_first_pivots_1003_8192x16384.json"


def _oracle_seconds_per_pivot(rows: int, cols: int, seed: int, pivots: int, blocked: bool = False):
    from dantzig_amd import core
    from oracle import oracle as ora

    a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
    sf = ora.stdform_from_dense(a, b, c)
    del a
    t0 = time.perf_counter()
    res = ora.simplex_solve(sf, max_iter=pivots, log_cap=pivots, blocked=blocked)
    dt = time.perf_counter() - t0
    return dt / max(res.iterations, 1), res.iterations, dt, res.pivots


CPU_AT_SIZE_MAX_ROWS = 8192


def cpu_baseline(sample_rows: int, sample_cols: int, seed: int, pivots: int, rows: int,
                 anchor_rows: int = 0, at_size: tuple | None = None):
    """The oracle (C restatement of the reference algorithm: full dense LU of B and of B^T
    every iteration) timed on THIS host on a bounded sample.

    `at_size` = (cols, seed, pivots): REAL pivots of the benchmark's own LP at its own size, by the
    oracle's blocked twin (oracle/dzg_oracle_blocked.c: Matrix::factorize applied block by block
    on several cores -- the same operations per element in the same order, its factors and pivots
    bit-equal to the literal loop's, tests/test_oracle_kats.py) on OMP_NUM_THREADS threads; the
    pivots it takes are compared with the committed fixture of the literal oracle.  `value` is then
    that measured rate and `cores` the threads.  The literal single-core loop is timed beside it on
    `pivots` pivots at `sample_rows` rows (and, with `anchor_rows`, one real pivot at that size, whose
    measured exponent carries the one-core rate to the benchmark size)."""
    import math

    spp, done, dt, _ = _oracle_seconds_per_pivot(sample_rows, sample_cols, seed, pivots)
    rate = 1.0 / spp
    out = {
        "unit": "iterations/s", "cores": 1, "kind": "port",
        "measured_value": rate, "measured_rows": sample_rows,
        "flop_model_value": rate * (sample_rows / rows) ** 3,
        "host_cores_total": os.cpu_count(),
    }
    sample = (f"first {done} pivots of the {sample_rows}x{sample_cols} G1 LP (seed {seed}) on the C "
              f"restatement of the reference (oracle/), one core, {dt:.1f} s of CPU: {rate:.3f} it/s")
    if anchor_rows and anchor_rows > sample_rows and rows >= anchor_rows:
        spp_a, _, dt_a, _ = _oracle_seconds_per_pivot(anchor_rows, 2 * anchor_rows, 1006, 1)
        expo = math.log(spp_a / spp) / math.log(anchor_rows / sample_rows)
        out["value"] = 1.0 / (spp_a * (rows / anchor_rows) ** expo)
        out["anchor"] = {"rows": anchor_rows, "seconds_per_pivot": spp_a, "pivots": 1,
                         "measured_exponent": expo}
        out["extrapolated"] = rows != anchor_rows
        sample += (f"; one real pivot at {anchor_rows} rows: {dt_a:.1f} s; the one-core rate at {rows} "
                   f"rows EXTRAPOLATED from that pivot with the exponent measured between the two "
                   f"sizes on this host ({expo:.2f}; the flop model (4/3)m^3 says 3 and would give "
                   f"{out['flop_model_value']:.4f} it/s)")
    else:
        out["value"] = out["flop_model_value"]
        out["extrapolated"] = rows != sample_rows
        sample += (f"; one-core rate at {rows} rows by the (4/3)m^3 flop model: that x "
                   f"({sample_rows}/{rows})^3 (no anchor pivot timed)")
    rec = None
    try:
        with open(os.path.join(ROOT, CPU_RECORD)) as f:
            rec = json.load(f)
        spp_rec = rec["seconds_per_pivot"]
        out["measured_at_benchmark_size"] = {
            "rows": rec["m"], "pivots": len(spp_rec),
            "seconds_per_pivot": sum(spp_rec) / len(spp_rec),
            "iterations_per_s": len(spp_rec) / sum(spp_rec),
            "where": "build container, 1 core (two other jobs on the machine), real pivots of the "
                     "8192x16384 seed-1003 LP", "source": CPU_RECORD}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        rec = None
    if at_size and rows <= CPU_AT_SIZE_MAX_ROWS:
        cols_s, seed_s, pivots_s = at_size
        spp_s, done_s, dt_s, log_s = _oracle_seconds_per_pivot(rows, cols_s, seed_s, pivots_s, blocked=True)
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        out["one_core"] = {"value": out["value"], "extrapolated": out["extrapolated"]}
        out["value"] = 1.0 / spp_s
        out["cores"] = threads
        out["extrapolated"] = False
        out["at_benchmark_size"] = {"rows": rows, "cols": cols_s, "seed": seed_s, "pivots": done_s,
                                    "seconds": dt_s, "seconds_per_pivot": spp_s, "threads": threads,
                                    "oracle": "blocked twin (oracle/dzg_oracle_blocked.c), bit-equal to the literal loop"}
        if rec and rec.get("m") == rows and rec.get("seed") == seed_s and rec.get("n_struct") == cols_s:
            n_cmp = min(done_s, len(rec["mu"]))
            want = [(int(rec["kind"][i]), int(rec["entering"][i]), int(rec["leaving"][i]), float(rec["mu"][i]))
                    for i in range(n_cmp)]
            got = [(int(p[0]), int(p[1]), int(p[2]), float(p[3])) for p in log_s[:n_cmp]]
            out["at_benchmark_size"]["pivots_equal_the_literal_oracles"] = n_cmp > 0 and got == want
        sample = (f"VALUE: {done_s} real pivots of the benchmark's own LP ({rows}x{cols_s}, seed {seed_s}) by "
                  f"the oracle's blocked twin on {threads} threads of this host, {dt_s:.1f} s: "
                  f"{1.0 / spp_s:.4f} it/s.  Beside it, one core: " + sample)
    out["sample"] = sample
    return out


def _run_in_own_group(cmd, cwd, env, timeout, stdout=None):
    """Runs a child in its own process group and, on a timeout, ends the WHOLE group: rocprofv3
    starts the profiled program as a grandchild, which would otherwise keep the GPU busy while
    this process goes on to the timed measurement."""
    import signal
    import subprocess

    proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=stdout or subprocess.DEVNULL,
                            stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        rc = proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        raise
    if rc != 0:
        raise RuntimeError(f"{cmd[0]} exited with {rc}")


def pmc_traffic(args, warm_k: int = 0, steps: int | None = None, warmup: int | None = None) -> dict | None:
    """HBM bytes per launch of the pricing kernel from the rocprofv3 PMC counters, collected in
    two separate child runs (FETCH_SIZE and WRITE_SIZE do not fit one pass) BEFORE this process
    touches the GPU.  Units are KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a
    wide coalesced streaming read (MI355X_MICROARCH.md, HBM), so the fetch side is doubled.
    warm_k > 0: the child starts from a basis of that many structural columns (the regime of the
    `deep` / `end` blocks without the pivots that lead there: the counters see every launch)."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    per_pass, same_run = {}, None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = tempfile.mkdtemp(prefix="dzg_pmc_", dir="/tmp")
            try:
                # the SAME pivots as the timed region (warm-up included: the counters see every
                # launch of the child), so that the bytes per pricing pass can stand beside the
                # algorithmic bytes of the same passes -- both depend on k, which grows with the pivots
                cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                       sys.executable, os.path.abspath(__file__), "--steps", str(steps),
                       "--warmup", str(warmup), "--warm-k", str(warm_k),
                       "--rows", str(args.rows), "--cols", str(args.cols), "--seed", str(args.seed),
                       "--price", args.price, "--no-cpu-baseline", "--no-pmc-traffic",
                       "--no-secondary", "--no-late"]
                if args.sparse_per_col > 0:
                    cmd += ["--sparse-per-col", str(args.sparse_per_col)]
                with open(os.path.join(out, "child.json"), "w") as child_out:
                    _run_in_own_group(cmd, "/tmp", dict(os.environ, TMPDIR="/tmp"), 300, child_out)
                with open(os.path.join(out, "child.json")) as f:
                    child = json.loads(f.read().strip().splitlines()[-1])
                passes = child["config"]["pivots_total"]
                same_run = child["config"]["price_bytes_total"] / max(passes, 1)
                total = 0.0
                for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"),
                                      recursive=True):
                    with open(path) as f:
                        for row in csv.DictReader(f):
                            if row["Counter_Name"] == counter and "k_price_" in row["Kernel_Name"]:
                                total += float(row["Counter_Value"])
            finally:
                shutil.rmtree(out, ignore_errors=True)
            if passes <= 0:
                return None
            per_pass[counter] = total / passes * 1024.0
    except Exception as exc:  # the profiler is optional: report null rather than fail the bench
        print(f"pmc traffic unavailable: {exc}", file=sys.stderr)
        return None
    fetch = 2.0 * per_pass["FETCH_SIZE"]
    return {"bytes_per_launch": fetch + per_pass["WRITE_SIZE"],
            "fetch_bytes_corrected": fetch, "write_bytes": per_pass["WRITE_SIZE"],
            "algorithmic_bytes_per_launch_same_pivots": same_run,
            "note": ("rocprofv3 --pmc, separate passes of the same pivots as the timed region, warm-up "
                     "included (all kernels of a pricing pass summed); FETCH_SIZE x2 (gfx950), KiB units"
                     if warm_k <= 0 else
                     "rocprofv3 --pmc, separate passes over %d pivots from a basis of %d structural columns "
                     "(factorised on the device: the regime of this block without the pivots that lead there; "
                     "all kernels of a pricing pass summed); FETCH_SIZE x2 (gfx950), KiB units"
                     % (steps + warmup, warm_k))}


def refactor_child(args) -> int:
    """--refactor-child K: create a solver of the workload warm-started from a basis of K structural
    columns -- dzg_solver_create factorises it on the device (csrc/k_refactor.hip) -- and exit.  Run
    under `rocprofv3 --pmc` by pmc_mfma() below; prints the seconds of a second, timed
    refactorisation (counters on: not a rate to quote)."""
    from dantzig_amd import core

    a, b, c = core.gen_dense_lp(seed=args.seed, m=args.rows, n_struct=args.cols)
    lp = core.warm_started(core.CoreLP.from_inequality_form(a, b, c), args.refactor_child)
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1) as s:
        t0 = time.perf_counter()
        s.refactor()
        print(json.dumps({"k": args.refactor_child, "refactor_s": time.perf_counter() - t0}))
    return 0


def pmc_mfma(args, k: int) -> dict | None:
    """MfmaUtil of the refactorisation: SQ_VALU_MFMA_BUSY_CYCLES over the SIMD cycles
    the kernels were resident, from a rocprofv3 --pmc child (started BEFORE this process touches
    the GPU) that factorises a basis of k structural columns twice (creation + one explicit call).
    The counters are summed over the 8 XCDs per dispatch; a kernel's duration in cycles is
    GRBM_GUI_ACTIVE / 8, the chip has 1024 SIMDs:  MfmaUtil = busy / (128 x GRBM_GUI_ACTIVE)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = tempfile.mkdtemp(prefix="dzg_mfma_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv",
               "-d", out, "--", sys.executable, os.path.abspath(__file__), "--refactor-child", str(k),
               "--rows", str(args.rows), "--cols", str(args.cols), "--seed", str(args.seed)]
        _run_in_own_group(cmd, "/tmp", dict(os.environ, TMPDIR="/tmp"), 300)
        per = {}
        for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if "k_ref_" not in name:
                        continue
                    short = name.split("(")[0].replace("void ", "").strip()
                    e = per.setdefault(short, {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "n": 0})
                    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    e["n"] += row["Counter_Name"] == "GRBM_GUI_ACTIVE"
        busy = sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] for e in per.values())
        active = sum(e["GRBM_GUI_ACTIVE"] for e in per.values())
        if active <= 0:
            return None
        gemm = {n: e for n, e in per.items() if e["SQ_VALU_MFMA_BUSY_CYCLES"] > 0}
        gb = sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] for e in gemm.values())
        ga = sum(e["GRBM_GUI_ACTIVE"] for e in gemm.values())
        return {"k": k, "MfmaUtil": busy / (128.0 * active),
                "MfmaUtil_inside_the_MFMA_kernels": gb / (128.0 * ga) if ga > 0 else None,
                "share_of_kernel_time_in_MFMA_kernels": ga / active,
                "kernels": {n: {"launches": e["n"], "MfmaUtil": e["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * e["GRBM_GUI_ACTIVE"])}
                            for n, e in sorted(gemm.items()) if e["GRBM_GUI_ACTIVE"] > 0},
                "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over every k_ref_* kernel of two "
                        "refactorisations of a basis of k structural columns (child run, same workload): busy "
                        "cycles / (128 x GRBM_GUI_ACTIVE) -- 1024 SIMDs, both counters summed over 8 XCDs"}
    except Exception as exc:  # the profiler is optional
        print(f"pmc mfma unavailable: {exc}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def refactor_at(rows, cols, seed, k: int) -> dict:
    """One timed dzg_solver_refactor of a basis of k structural columns of the workload (the LP
    warm-started there, unprofiled): seconds and TFLOP/s by the flop model of the `mfma` block."""
    from dantzig_amd import core

    a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
    lp = core.warm_started(core.CoreLP.from_inequality_form(a, b, c), k)
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1) as s:
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            s.refactor()
            best = min(best, time.perf_counter() - t0)
    flops = 2.0 * k ** 3 + 2.0 * (rows - k) * k ** 2
    return {"k": k, "seconds": best, "achieved": flops / best / 1e12, "unit": "TFLOP/s",
            "frac": flops / best / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "frac_of_datasheet": flops / best / 1e12 / MFMA_F64_DATASHEET_TFLOPS,
            "note": "best of three dzg_solver_refactor calls, wall clock around the call"}


SECONDARY = {"rows": 32768, "cols": 65536, "seed": 1005, "steps": 300, "warmup": 50}
# the same LP deep in its solve: warm-started from a basis of 16 384 structural columns (the compact
# inverse 4.3 GB, a row-wise pricing pass 8.6 GB) -- the state the column- and row-sharded solve of
# `bench.py --gpus N` is measured on as well ("deep" block there)
SECONDARY_DEEP = {"warm_k": 16384, "steps": 300, "warmup": 50}


def under_profiler() -> bool:
    """True when this process was started by rocprofv3 (its tool library is preloaded and has
    already touched the GPU).  The run is then kept to the primary workload in this one process:
    no child processes (none may be exec'ed once the GPU is initialised; nested profilers do not
    mix) and no second workload, so the profiler's per-kernel averages are those of `value`."""
    return ("ROCP_TOOL_LIBRARIES" in os.environ
            or "rocprofiler" in os.environ.get("LD_PRELOAD", ""))


def secondary_wanted(args) -> bool:
    """The config-5 measurement rides along with the default invocation only."""
    return (not args.no_secondary and not under_profiler() and args.rows == 8192
            and args.cols == 16384 and args.sparse_per_col == 0 and args.numerics == "fast")


EVENT_STRIDE = 8  # pricing passes per timed one (opts.profile bits 16..23)


def price_profile() -> int:
    from dantzig_amd import _ffi

    return (1 << _ffi.K_PRICE) | (EVENT_STRIDE << 16)


PRICE_KERNELS = {"auto": "k_price_tree", "tree": "k_price_tree", "seq": "k_price_seq2",
                 "wave": "k_price_wave2"}


ROWS_T = None  # set by measure(): the compact width below which the dense pass runs row-wise
ROWS_M = 0


def _pricing(r0, r1, kernel: str) -> dict:
    """Roofline block of the pricing kernel between two result snapshots."""
    ms = r1.kernel_ms["price"] - r0.kernel_ms["price"]
    timed = r1.kernel_launches["price"] - r0.kernel_launches["price"]
    launches = r1.iterations - r0.iterations  # one pricing pass per pivot
    nbytes = r1.price_bytes - r0.price_bytes  # (algorithmic, every pass of the region)
    avg_ms = ms / max(timed, 1)
    per_launch = nbytes / max(launches, 1)
    achieved = (per_launch / 1e9) / (avg_ms / 1e3) if ms > 0 else float("nan")
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "avg_launch_us": 1e3 * avg_ms, "launches": launches, "launches_timed": timed,
           "timing": ("HIP events (no system-scope fence) around every %d-th pricing pass of the region: "
                      "an event pair between two 10-us kernels costs the GPU a few idle microseconds, "
                      "5.7 each with default events (profiles/r04_chain_phase_clocks.txt)" % EVENT_STRIDE),
           "algorithmic_bytes_per_launch": per_launch}
    if ROWS_T is not None:
        # dense matrix, AUTO pricing: the pass is chosen per pivot by the compact width k
        k0, k1 = r0.dense_columns, r1.dense_columns
        small_k = int(os.environ.get("DZG_PRICE_SMALL_K", "480"))
        if max(k0, k1) + 50 < small_k:  # (the engine's bound on k for a batch of 50 pivots)
            out["kernel"] = "k_price_rows_small"
            out["bytes_model"] = ("fused row-wise pass: 8 (k+1) ldt (rows of the row-major copy) + 4 n_s (the "
                                  "code -> position map) + 12 k + 32 q per launch (the row groups' partial sums "
                                  "stay in LDS); k = %d..%d of %d rows" % (k0, k1, ROWS_M))
        elif max(k0, k1) < ROWS_T:
            out["kernel"] = "k_price_rows<2> + k_price_rows_finish"
            out["bytes_model"] = ("row-wise pass: 8 (k+1) ldt (rows of the row-major copy) + 16 G ldt "
                                  "(partials of the G row groups, written and read) + 12 k + 32 q per "
                                  "launch pair; k = %d..%d of %d rows" % (k0, k1, ROWS_M))
        elif min(k0, k1) >= ROWS_T:
            out["kernel"] = "k_price_tree"
        else:
            out["kernel"] = "k_price_rows<2> + k_price_rows_finish below k = %d, k_price_tree from there" % ROWS_T
        out["rows_T"] = ROWS_T
    return out


def _late_regime(solver, r1, steps: int, late_pivots: int, kernel: str) -> dict:
    """The same solve deep into its trajectory: `late_pivots` pivots are skipped untimed (the
    basis inverse has grown to k dense columns by then, so FTRAN, the eta flush and the update
    cost what they cost for most of a whole solve), then max(`steps`, 1000) pivots are timed like
    the primary region, then 256 more with every kernel class stamped for the per-class split."""
    from dantzig_amd import _ffi

    skip = late_pivots - r1.iterations
    status = solver.run(skip) if skip > 0 else "iter_limit"
    if status != "iter_limit":
        return {"skipped": f"solve ended ({status}) before pivot {late_pivots}"}
    steps = max(steps, 1000)  # a short --steps would time one poll batch: too noisy to compare
    ra = solver.result(log=False)
    t0 = time.perf_counter()
    status = solver.run(steps)
    elapsed = time.perf_counter() - t0
    rb = solver.result(log=False)
    done = rb.iterations - ra.iterations
    out = {"untimed_pivots_before": ra.iterations, "steps": done,
           "value": done / elapsed if elapsed > 0 else float("nan"), "unit": "iterations/s",
           "ms_per_step": 1e3 * elapsed / max(done, 1),
           "k_at_start": ra.dense_columns, "k_at_end": rb.dense_columns,
           "max_pivot_error": rb.max_pivot_error, "near_ties": rb.near_ties,
           "first_near_tie": rb.first_near_tie,
           "refactors": rb.refactors, "roofline": _pricing(ra, rb, kernel)}
    if status == "iter_limit":
        solver.set_profile((1 << _ffi.K_COUNT) - 1)
        solver.run(256)
        rc = solver.result(log=False)
        n = max(rc.iterations - rb.iterations, 1)
        if rc.kernel_launches["ratio"] > rb.kernel_launches["ratio"] and \
                rc.kernel_launches["status"] == rb.kernel_launches["status"]:
            # the sparse-basis path (csrc/k_sparse.hip): status() is the head of the FTRAN launch
            if rc.kernel_launches["btran"] > rb.kernel_launches["btran"]:  # eight launches (DZG_SP_FUSED=0)
                names = {"ftran": "status + primal FTRAN (k_sp_ftran_s<0>, k_sp_ftran_l)",
                         "btran": "BTRAN row", "price": "pricing",
                         "ratio": "dual ratio test + FTRAN (k_sp_ftran_s<1>, k_sp_ftran_l)",
                         "update": "pivot + update", "basis_update": "eta flush (amortised)"}
            else:  # four launches
                names = {"ftran": "k_sp_pre: status, primal FTRAN (both halves) + ratio test, BTRAN row",
                         "price": "pricing (k_price_csc_rl)",
                         "ratio": "k_sp_mid: dual ratio test + FTRAN (both halves), the pivot's books",
                         "update": "k_sp_update", "basis_update": "eta flush (amortised)"}
        elif rc.kernel_launches["status"] > rb.kernel_launches["status"]:
            names = {"status": "status + primal FTRAN prep", "ftran": "FTRAN GEMV (primal)",
                     "btran": "BTRAN row", "price": "pricing",
                     "ratio": "dual ratio + prep + FTRAN GEMV", "update": "pivot + update",
                     "basis_update": "eta flush (amortised)"}
        else:  # the three-launch chain (csrc/k_chain.hip)
            names = {"ftran": "k_chain_pre: status, primal FTRAN + ratio test, BTRAN row",
                     "price": "pricing",
                     "update": "k_chain_post: dual ratio test + FTRAN, pivot's books, update",
                     "basis_update": "eta flush (amortised)"}
        out["kernel_us_per_pivot"] = {
            label: round(1e3 * (rc.kernel_ms[k] - rb.kernel_ms[k]) / n, 2)
            for k, label in names.items()}
        out["kernel_us_note"] = (f"HIP events around each kernel class over {n} further pivots "
                                 "(event overhead included; not part of any reported rate)")
        solver.set_profile(price_profile())  # back to timing the pricing pass only
    return out


SEVEN_LAUNCHES = False  # --seven-launches: the FAST iteration as seven kernels instead of three


def measure(rows, cols, seed, sparse_per_col, price_name, numerics_name, steps, warmup,
            late_pivots: int = 0, deep_pivots: int = 0, whole_solve: bool = False,
            mfma_block: bool = False, end_pivots: int = 0, warm_k: int = 0) -> dict:
    """One workload on one GPU: generate, upload (untimed), `warmup` pivots, then `steps` timed;
    with late_pivots > 0 a second timed region deep in the same solve (see _late_regime)."""
    from dantzig_amd import _ffi, core

    t_gen = time.perf_counter()
    if sparse_per_col > 0:
        cp, ri, val, b, c = core.gen_sparse_lp(seed, rows, cols, sparse_per_col)
        lp = core.CoreLP.from_csc(rows, cp, ri, val, b, c)
    else:
        a, b, c = core.gen_dense_lp(seed=seed, m=rows, n_struct=cols)
        lp = core.CoreLP.from_inequality_form(a, b, c)
        if warm_k > 0:  # (the deep regime of the solve without the pivots that lead there)
            lp = core.warm_started(lp, warm_k)
    t_gen = time.perf_counter() - t_gen
    price = {"auto": core.PRICE_AUTO, "seq": core.PRICE_SEQ, "wave": core.PRICE_WAVE,
             "tree": core.PRICE_TREE}[price_name]
    numerics = core.FAST if numerics_name == "fast" else core.STRICT
    # (one GPU, FAST: the sparse-basis path prices over the live entries of a column only;
    # DZG_SP_PRICE_FULL=1 brings the full pass back for comparison)
    # dense, AUTO pricing: row-wise while k < rows_T (csrc/k_price_kernels.h), column-wise beyond
    rows_T = int(0.93 * rows * cols / (rows + cols))
    if os.environ.get("DZG_PRICE_ROWS_T"):
        rows_T = int(os.environ["DZG_PRICE_ROWS_T"])
    rows_on = (sparse_per_col <= 0 and numerics_name == "fast" and price_name == "auto"
               and os.environ.get("DZG_PRICE_ROWS", "1") != "0")
    global ROWS_T, ROWS_M
    ROWS_T, ROWS_M = (rows_T if rows_on else None), rows
    kernel = (("k_price_csc_tree" if os.environ.get("DZG_SP_PRICE_FULL") == "1" else "k_price_csc_rl")
              if numerics_name == "fast" and price_name != "seq"
              else "k_price_csc") if sparse_per_col > 0 else (
        "k_price_seq2" if numerics_name == "strict" else PRICE_KERNELS[price_name])
    t_up = time.perf_counter()
    solver = core.Solver(lp, numerics=numerics, price_kernel=price,
                         profile=price_profile(), poll_interval=50,
                         seven_launches=1 if SEVEN_LAUNCHES else 0)
    t_up = time.perf_counter() - t_up
    late = deep = whole = mfma = end = None
    try:
        status = solver.run(warmup) if warmup > 0 else "iter_limit"
        r0 = solver.result(log=False)
        t0 = time.perf_counter()
        if status == "iter_limit":
            status = solver.run(steps)
        t1 = time.perf_counter()
        r1 = solver.result(log=False)
        if late_pivots > 0 and status == "iter_limit" and numerics_name == "fast":
            late = _late_regime(solver, r1, steps, late_pivots, kernel)
            if deep_pivots > late_pivots and "skipped" not in late:
                # the same solve deeper still: the basis inverse is several thousand columns wide
                # and FTRAN streams as many bytes as the pricing pass has shed
                deep = _late_regime(solver, solver.result(log=False), steps, deep_pivots, kernel)
        if mfma_block and numerics_name == "fast" and sparse_per_col <= 0 and status == "iter_limit":
            # the refactorisation's GEMM side against the fp64 MFMA peak -- ONE timed
            # dzg_solver_refactor at the deepest regime reached above (blocked LU with partial
            # pivoting of the k x k structural block, the explicit inverse, the slack rows: all but
            # the panels are MFMA GEMMs, csrc/k_refactor.hip).  After every timed region.
            rk = solver.result(log=False)
            k = rk.dense_columns
            if k >= 256:
                tr = time.perf_counter()
                solver.refactor()
                tr = time.perf_counter() - tr
                flops = 2.0 * k ** 3 + 2.0 * (rows - k) * k ** 2
                mfma = {"bound": "mfma", "what": "dzg_solver_refactor (k_refactor.hip) at k = %d of m = %d" % (k, rows),
                        "k": k, "seconds": tr, "flops_model": "2 k^3 (LU + forward and backward substitution of the identity) + 2 (m - k) k^2 (slack rows)",
                        "achieved": flops / tr / 1e12, "unit": "TFLOP/s", "peak": MFMA_F64_PEAK_TFLOPS,
                        "frac": flops / tr / 1e12 / MFMA_F64_PEAK_TFLOPS,
                        "peak_note": "v_mfma_f64_16x16x4_f64 measured on this chip by tools/mfma_f64_peak.hip "
                                     "(profiles/r02_mfma_f64_peak_microbench.txt); the datasheet's 78.6 is not "
                                     "reached by that instruction",
                        "frac_of_datasheet": flops / tr / 1e12 / MFMA_F64_DATASHEET_TFLOPS,
                        "datasheet_peak": MFMA_F64_DATASHEET_TFLOPS,
                        "MfmaUtil": None}  # (filled by main() from the rocprofv3 --pmc child)
        if end_pivots > 0 and numerics_name == "fast" and sparse_per_col <= 0 and \
                solver.result(log=False).status in ("iter_limit", "running"):
            # the regime that owns the whole solve: k beyond rows_T, the pricing pass column-wise
            # again and FTRAN streaming 8 m k bytes of the compact inverse
            end = _late_regime(solver, solver.result(log=False), steps, end_pivots, kernel)
        if whole_solve and numerics_name == "fast":
            # the rest of the solve, to optimality (the untimed regime blocks above are part of the
            # same trajectory; pivots and seconds are those of this last stretch)
            ra = solver.result(log=False)
            tw = time.perf_counter()
            wstatus = solver.run(0)
            tw = time.perf_counter() - tw
            rb = solver.result(log=False)
            whole = {"status": wstatus, "pivots_total": rb.iterations,
                     "pivots_timed": rb.iterations - ra.iterations, "seconds_timed": round(tw, 2),
                     "value": (rb.iterations - ra.iterations) / tw if tw > 0 else float("nan"),
                     "unit": "iterations/s", "k_at_end": rb.dense_columns, "objective": rb.objective,
                     "max_pivot_error": rb.max_pivot_error, "near_ties": rb.near_ties,
                     "first_near_tie": rb.first_near_tie, "state_drift": rb.state_drift,
                     "price_pass_used": rb.price_pass_used,
                     "refactors": rb.refactors, "chain_fallbacks": rb.chain_fallbacks,
                     # the whole solve as one number: every pivot from the first over the time spent
                     # inside dzg_solver_run (the regime blocks above are stretches of this solve)
                     "whole_solve": {"pivots": rb.iterations, "seconds": round(rb.solve_ms / 1e3, 2),
                                     "value": rb.iterations / (rb.solve_ms / 1e3) if rb.solve_ms > 0 else float("nan"),
                                     "unit": "iterations/s"}}
    finally:
        solver.close()

    steps_done = r1.iterations - r0.iterations
    elapsed = t1 - t0
    out = {
        "metric": "simplex_iterations_per_sec",
        "value": steps_done / elapsed if elapsed > 0 else float("nan"),
        "unit": "iterations/s",
        "n_gpus": 1,
        "steps": steps_done,
        "warmup": r0.iterations,
        "ms_per_step": 1e3 * elapsed / max(steps_done, 1),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"dense random LP {rows}x{cols} fp64, generator G1 seed {seed}"
                         if sparse_per_col <= 0 else
                         f"sparse random LP {rows}x{cols} fp64, {sparse_per_col} "
                         f"nonzeros per column (CSC), generator G2 seed {seed}"),
            "numerics": r1.numerics,
            "price_kernel": price_name,
            "launches_per_iteration": ((8 if os.environ.get("DZG_SP_FUSED") == "0" else 4)
                                       if sparse_per_col > 0 else 7 if SEVEN_LAUNCHES else
                                       4 if (rows_on and r1.dense_columns < rows_T) else 3)
            if numerics_name == "fast" else None,
            "status_after_timed_region": status,
            "requested_steps": steps,
            "k_at_start": r0.dense_columns,
            "k_at_end": r1.dense_columns,
            "pivots_total": r1.iterations,
            "price_bytes_total": r1.price_bytes,
            "lp_generation_s": round(t_gen, 3),
            "upload_s": round(t_up, 3),
            "max_pivot_error": r1.max_pivot_error,
            "near_ties": r1.near_ties,
            "first_near_tie": r1.first_near_tie,
            "price_pass_used": {0: "none", 1: "rows", 2: "columns", 3: "rows then columns"}[r1.price_pass_used & 3],
            "price_rows_copy": bool(r1.price_rows_copy),
            "refactors": r1.refactors,
        },
        "roofline": _pricing(r0, r1, kernel),
    }
    if late is not None:
        out["late"] = late
    if deep is not None:
        out["deep"] = deep
    if end is not None:
        out["end"] = end
    if mfma is not None:
        out["mfma"] = mfma
    if whole is not None:
        out["whole_solve_remainder"] = whole
        out["whole_solve"] = whole["whole_solve"]  # (the same figure where a reader looks first)
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--seed", type=int, default=1003)
    ap.add_argument("--price", choices=["auto", "tree", "seq", "wave"], default="auto")
    ap.add_argument("--numerics", choices=["fast", "strict"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc-traffic", action="store_true")
    ap.add_argument("--no-late", action="store_true",
                    help="skip the second timed region deep in the solve (the \"late\" block)")
    ap.add_argument("--late-pivots", type=int, default=20000,
                    help="pivots skipped untimed before the late region (default 20000, ~4 s)")
    ap.add_argument("--deep-pivots", type=int, default=150000,
                    help="a third timed region after this many pivots (default 150000: k ~ 4000 on the "
                         "benchmark LP, ~30 s untimed); 0 or --no-late = skip")
    ap.add_argument("--end-pivots", type=int, default=400000,
                    help="a fourth timed region after this many pivots (default 400000: k ~ 7700 on the "
                         "benchmark LP, the regime most of the solve's time is spent in); rides with the "
                         "whole solve of the default workload, 0 = skip")
    ap.add_argument("--refactor-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-mfma", action="store_true",
                    help="skip the timed refactorisation (the \"mfma\" block) after the timed regions")
    ap.add_argument("--whole-solve", action="store_true",
                    help="after the timed regions run the solve to its end and report the stretch and "
                         "the whole solve's average (the benchmark LP: ~515 000 pivots, ~1.5 minutes); "
                         "on by default for the default workload")
    ap.add_argument("--no-whole-solve", action="store_true")
    ap.add_argument("--seven-launches", action="store_true",
                    help="FAST, dense, one GPU: run an iteration as the seven launches a sharded "
                         "solver uses instead of the three-launch chain (same pivots)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the 32768x65536 measurement that rides along with the default run")
    ap.add_argument("--sparse-per-col", type=int, default=0,
                    help="generator G2: this many nonzeros per column, matrix kept CSC on the device")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the column-sharded RCCL path even with one rank (rehearsal)")
    ap.add_argument("--warm-k", type=int, default=0,
                    help="start from a basis of this many structural columns (x = 1, z = -1; factorised on the "
                         "device before the warm-up) instead of the slack basis: a regime of the solve without "
                         "the pivots that lead there")
    ap.add_argument("--cpu-sample-rows", type=int, default=1024)
    ap.add_argument("--cpu-sample-pivots", type=int, default=100)
    ap.add_argument("--cpu-anchor-rows", type=int, default=0,
                    help="CPU baseline: also time ONE real oracle pivot at this many rows on one core "
                         "(~20 s at 4096) to anchor the one-core extrapolation to the benchmark size; 0 = skip")
    ap.add_argument("--cpu-size-pivots", type=int, default=6,
                    help="CPU baseline: real pivots of the benchmark's own LP at its own size by the oracle's "
                         "blocked twin on 16 threads (3.4 s each at 8192 rows on a GPU box); 0 = skip")
    args = ap.parse_args()

    if args.refactor_child > 0:
        return refactor_child(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        print(f"--gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
        return 2
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        # asked for several GPUs without a launcher: start one rank per GPU ourselves (this process
        # has not touched the GPU yet) and pass the child's JSON line through
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.abspath(__file__), *sys.argv[1:]]
        return subprocess.run(cmd).returncode
    if world > 1 or args.force_sharded:
        from dantzig_amd import sharded  # column-sharded path, native RCCL loop

        return sharded.bench_main(args, rank, world, local_rank)

    # children first: no GPU state yet (and none at all under a profiler)
    traffic = None if (args.no_pmc_traffic or under_profiler()) else pmc_traffic(args)
    default_dense = args.rows == 8192 and args.cols == 16384 and args.sparse_per_col == 0 \
        and args.numerics == "fast"
    # the same counters where the pass STREAMS: children warm-started at the compact widths of the
    # `deep` and `end` blocks (default invocation only)
    regime_traffic = {}
    if traffic is not None and default_dense and not (args.no_late or args.warm_k):
        for name, wk in (("deep", 4050), ("end", 7650)):
            regime_traffic[name] = pmc_traffic(args, warm_k=wk, steps=200, warmup=50)
    mfma_wanted = default_dense and not (args.no_late or args.no_mfma or under_profiler())
    mfma_pmc = pmc_mfma(args, args.rows) if (mfma_wanted and not args.no_pmc_traffic) else None

    from dantzig_amd import _ffi

    # one HIP runtime per process: the library's.  dzg_solver_run returns after synchronising
    # its stream, which brackets the timed region the way torch.cuda.synchronize() would.
    _ffi.require_gpu()
    late_pivots = 0 if (args.no_late or under_profiler()) else args.late_pivots
    global SEVEN_LAUNCHES
    SEVEN_LAUNCHES = bool(args.seven_launches)
    global EVENT_STRIDE  # (a short timed region still gets a few timed passes)
    EVENT_STRIDE = 8 if args.steps >= 64 else (4 if args.steps >= 16 else 1)
    deep_pivots = args.deep_pivots if (late_pivots > 0 and args.rows == 8192 and args.cols == 16384
                                       and args.sparse_per_col == 0) else 0
    whole = (args.whole_solve or (deep_pivots > 0 and args.numerics == "fast")) and not args.no_whole_solve
    out = measure(args.rows, args.cols, args.seed, args.sparse_per_col, args.price, args.numerics,
                  args.steps, args.warmup, late_pivots, deep_pivots, whole,
                  mfma_block=late_pivots > 0 and not args.no_mfma,
                  end_pivots=args.end_pivots if (whole and deep_pivots > 0) else 0, warm_k=args.warm_k)
    if "mfma" in out and mfma_wanted:
        # config 3 is named after this: the refactorisation of the FULL basis (k = m = 8192), timed
        # unprofiled, and its MFMA utilisation from the counters of the child run above
        out["mfma"]["full_basis"] = refactor_at(args.rows, args.cols, args.seed, args.rows)
        out["mfma"]["MfmaUtil"] = mfma_pmc["MfmaUtil"] if mfma_pmc else None
        out["mfma"]["MfmaUtil_detail"] = mfma_pmc
    out["roofline"]["traffic"] = traffic["bytes_per_launch"] if traffic else None
    out["roofline"]["traffic_detail"] = traffic
    for name, tr in regime_traffic.items():
        if tr and name in out and "roofline" in out[name]:
            out[name]["roofline"]["traffic"] = tr["bytes_per_launch"]
            out[name]["roofline"]["traffic_detail"] = tr
    if ROWS_T is not None:
        # the pass moves 8 (k+1) n_s bytes: launch-bound while k is a few hundred, a stream later in
        # the same solve -- the fractions measured there, side by side (blocks "late" and "deep")
        regimes = {"k=%d..%d (timed region)" % (out["config"]["k_at_start"], out["config"]["k_at_end"]):
                   out["roofline"]["frac"]}
        for name in ("late", "deep", "end"):
            blk = out.get(name)
            if blk and "roofline" in blk:
                regimes["k=%d..%d (%s)" % (blk["k_at_start"], blk["k_at_end"], name)] = blk["roofline"]["frac"]
        out["roofline"]["frac_by_regime"] = regimes
        out["roofline"]["frac_note"] = ("the column-wise kernel (k >= rows_T, and every pivot before the "
                                        "row-wise pass existed) streams at 0.80-0.83 of the HBM peak "
                                        "(profiles/r03_bench_before_row_pricing.json)")
    if under_profiler():
        out["config"]["under_profiler"] = True
    if secondary_wanted(args):
        # the LP the north star's multi-GPU target is quoted on (config 5), on this one GPU too, so
        # that the per-N lines of a scaling run can be compared on it as well as on config 3
        try:
            sec = measure(SECONDARY["rows"], SECONDARY["cols"], SECONDARY["seed"], 0, "auto", "fast",
                          SECONDARY["steps"], SECONDARY["warmup"])
            out["secondary"] = {k: sec[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step",
                                                    "config", "roofline")}
            dp = SECONDARY_DEEP
            sec = measure(SECONDARY["rows"], SECONDARY["cols"], SECONDARY["seed"], 0, "auto", "fast",
                          dp["steps"], dp["warmup"], warm_k=dp["warm_k"])
            out["secondary"]["deep"] = {k: sec[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step",
                                                            "config", "roofline")}
            out["secondary"]["deep"]["note"] = (
                "the same LP warm-started from a basis of %d structural columns (x = 1, z = -1; factorised "
                "on the device before the warm-up): the one-GPU figure the `deep` block of `bench.py --gpus N` "
                "stands beside" % dp["warm_k"])
        except Exception as exc:  # never lose the primary line to the secondary workload
            out.setdefault("secondary", {})["error"] = f"{type(exc).__name__}: {exc}"
    if not args.no_cpu_baseline:
        at_size = (args.cols, args.seed, args.cpu_size_pivots) if args.cpu_size_pivots > 0 and not args.sparse_per_col else None
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_rows, 2 * args.cpu_sample_rows, 1002,
                                           args.cpu_sample_pivots, args.rows, args.cpu_anchor_rows, at_size)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())

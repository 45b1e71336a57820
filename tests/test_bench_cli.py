"""bench.py host logic that needs no GPU: what rides along with which invocation, profiler
detection, and the loud failure when there is no device."""
import argparse
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(**kw):
    base = dict(rows=8192, cols=16384, sparse_per_col=0, numerics="fast", no_secondary=False)
    base.update(kw)
    return argparse.Namespace(**base)


def test_secondary_workload_rides_only_with_the_default_invocation(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES", raising=False)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert bench.secondary_wanted(_args())
    assert not bench.secondary_wanted(_args(no_secondary=True))
    assert not bench.secondary_wanted(_args(rows=1024, cols=2048))
    assert not bench.secondary_wanted(_args(sparse_per_col=50))
    assert not bench.secondary_wanted(_args(numerics="strict"))
    assert bench.SECONDARY["rows"] == 32768 and bench.SECONDARY["cols"] == 65536


def test_profiler_detection_keeps_the_run_to_one_workload(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES", raising=False)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler() and not bench.secondary_wanted(_args())
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk.so")
    assert bench.under_profiler()


def test_bench_fails_loudly_without_a_gpu():
    from dantzig_amd import _ffi

    if _ffi.lib().dzg_device_count() > 0:
        import pytest

        pytest.skip("a GPU is visible")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-pmc-traffic", "--no-cpu-baseline", "--no-secondary"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode != 0
    assert "no HIP device" in run.stderr or "no CPU" in run.stderr
    assert run.stdout.strip() == ""  # no JSON line is ever printed for a run that measured nothing


def test_profiler_child_is_killed_with_its_whole_process_group(tmp_path):
    """pmc_traffic's children run in their own process group and a timeout ends the group: the
    profiled program is rocprofv3's grandchild and must not outlive it on the GPU."""
    import time

    bench = _bench()
    marker = tmp_path / "grandchild.pid"
    script = ("import os, subprocess, sys, time\n"
              f"p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)'])\n"
              f"open({str(marker)!r}, 'w').write(str(p.pid))\n"
              "time.sleep(60)\n")
    t0 = time.time()
    try:
        bench._run_in_own_group([sys.executable, "-c", script], str(tmp_path), dict(os.environ), 2)
        raise AssertionError("expected a timeout")
    except subprocess.TimeoutExpired:
        pass
    assert time.time() - t0 < 30
    pid = int(marker.read_text())
    for _ in range(50):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the grandchild survived the timeout")


def test_cpu_baseline_measures_the_benchmarks_own_lp_when_asked(monkeypatch):
    """cpu_baseline: `value` is the oracle's blocked twin on real pivots of the benchmark's own LP at
    its own size (threads stated), the literal one-core sample stays beside it; without the at-size
    leg the value is the one-core figure carried to the benchmark size.  (Small sizes here: the
    structure of the block, not its numbers.)"""
    monkeypatch.setenv("OMP_NUM_THREADS", "2")
    bench = _bench()
    both = bench.cpu_baseline(128, 256, 1002, 10, 256, 0, (512, 1003, 3))
    assert both["kind"] == "port" and both["cores"] == 2 and both["extrapolated"] is False
    at = both["at_benchmark_size"]
    assert (at["rows"], at["cols"], at["seed"], at["pivots"], at["threads"]) == (256, 512, 1003, 3, 2)
    assert abs(both["value"] - 1.0 / at["seconds_per_pivot"]) <= 1e-12 * both["value"]
    assert both["one_core"]["extrapolated"] is True and both["measured_rows"] == 128
    assert "pivots_equal_the_literal_oracles" not in at  # (no committed fixture for this LP)
    alone = bench.cpu_baseline(128, 256, 1002, 10, 256, 0, None)
    assert alone["cores"] == 1 and alone["extrapolated"] is True and "at_benchmark_size" not in alone
    # beyond 8192 rows a pivot of the reference algorithm takes minutes on any host: no at-size leg
    big = bench.cpu_baseline(128, 256, 1002, 10, 32768, 0, (65536, 1005, 2))
    assert "at_benchmark_size" not in big and big["cores"] == 1


def test_event_stride_rides_in_the_profile_word():
    bench = _bench()
    from dantzig_amd import _ffi

    bench.EVENT_STRIDE = 8
    assert bench.price_profile() == (1 << _ffi.K_PRICE) | (8 << 16)

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

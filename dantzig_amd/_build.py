"""Builds dantzig_amd/libdantzig_amd.so for gfx950 with hipcc (in-tree, no JIT cache).

-ffp-contract=off is deliberate and global: hipcc contracts a*b+c into FMA on the device by
default, the reference (Rust f64) never does.  Kernels that want an FMA
call fma() explicitly.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libdantzig_amd.so")
ARCH = "gfx950"
SOURCES = ["engine.hip", "k_vector.hip", "k_price.hip", "k_strict.hip", "k_fast.hip", "k_chain.hip", "k_sparse.hip", "k_refactor.hip", "k_rowshard.hip", "k_drift.hip",
           "model.cpp", "lpgen.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: dantzig_amd needs the ROCm toolchain to build")
    return exe


def _newest(paths) -> float:
    return max(os.path.getmtime(p) for p in paths)


def _deps():
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "k_price_kernels.h"),
            os.path.join(CSRC, "fast_decide.h"), os.path.join(CSRC, "fast_rows.h"),
            os.path.join(CSRC, "chain_barrier.h"),
            os.path.join(os.path.dirname(HERE), "include", "dantzig_amd.h")]
    return hdrs


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= _newest([path] + _deps()):
        return obj
    cmd = [_hipcc(), f"--offload-arch={ARCH}", *FLAGS, "-c", path, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "c++")
        cmd = [c for c in cmd if not c.startswith("--offload-arch")]
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-pthread"]
        subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    import sys

    build(force="--force" in sys.argv, verbose=True)

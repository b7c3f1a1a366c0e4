"""Builds libsbr_hip.so (the gfx950 engine) in-tree with hipcc.

    python -m sbr_rs_amd.build            # incremental
    python -m sbr_rs_amd.build --force

Flags that are part of the numerics contract (sbr_numerics.h): -ffp-contract=off (every fused
multiply-add is written explicitly), no -ffast-math, default (IEEE) f32 division/sqrt and
denormal handling.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsbr_hip.so")
SOURCES = ["sbr_kernels.hip", "sbr_steps.hip", "sbr_sort.hip", "sbr_wave.hip", "sbr_report.hip", "sbr_engine.hip"]
HEADERS = ["sbr_kernels.h", "sbr_device.h", "sbr_wave_seq.h", "sbr_numerics.h", "sbr_approx.h", "sbr_ziggurat_tables.h", os.path.join("..", "..", "include", "sbr_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("SBR_EXTRA_FLAGS", "").split()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    cc = hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + hdrs):
            cmd = [cc] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


REPO = os.path.dirname(HERE)
FACADE_SRC = os.path.join(REPO, "tests", "cpp", "facade_tests.cpp")
FACADE_BIN = os.path.join(REPO, "tests", "cpp", "_build", "facade_tests")


def build_facade_tests(force: bool = False, verbose: bool = True) -> str:
    """g++ build of the C++ host layer's test program (include/sbr.hpp over libsbr_hip.so)."""
    cxx = shutil.which("g++") or "g++"
    deps = [FACADE_SRC, os.path.join(REPO, "include", "sbr.hpp"), os.path.join(REPO, "include", "sbr_hip.h"), LIB]
    if force or _stale(FACADE_BIN, deps):
        os.makedirs(os.path.dirname(FACADE_BIN), exist_ok=True)
        cmd = [cxx, "-std=c++17", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(REPO, "include"), FACADE_SRC, "-o", FACADE_BIN,
               "-L" + HERE, "-lsbr_hip", "-Wl,-rpath,$ORIGIN/../../../sbr_rs_amd"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return FACADE_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_facade_tests(force="--force" in sys.argv))

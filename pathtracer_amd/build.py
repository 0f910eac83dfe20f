"""Builds the native libraries in-tree: lib/libptx355.so (HIP kernels + C ABI, gfx950) and
lib/libpthost.so (host-side scene ingest, plain C++)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "lib")
CSRC = os.path.join(HERE, "csrc")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
HOST_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-pthread"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(*dirs):
    out = []
    for d in dirs:
        for root, _, files in os.walk(d):
            out += [os.path.join(root, f) for f in files if f.endswith((".h", ".hip", ".cpp"))]
    return out


def build_device(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libptx355.so")
    deps = _deps(CSRC, os.path.join(os.path.dirname(HERE), "include"))
    if force or _stale(target, deps):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        cmd = [hipcc] + HIPCC_FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + \
              ["-o", target, os.path.join(CSRC, "ptx355.hip")]
        subprocess.check_call(cmd)
    return target


def build_host(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libpthost.so")
    deps = _deps(os.path.join(CSRC, "host"), os.path.join(os.path.dirname(HERE), "include"))
    if force or _stale(target, deps):
        subprocess.check_call(["g++"] + HOST_FLAGS + ["-o", target, os.path.join(CSRC, "host", "pth_host.cpp")])
    return target


def build_harness(force: bool = False) -> str:
    """tools/pt355.cpp: the C++ host over both C ABIs (bin/pt355, finds the libraries next to it through its rpath)."""
    bindir = os.path.join(HERE, "bin")
    os.makedirs(bindir, exist_ok=True)
    target = os.path.join(bindir, "pt355")
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "pt355.cpp")
    deps = [src, os.path.join(LIB, "libptx355.so"), os.path.join(LIB, "libpthost.so")] + _deps(os.path.join(root, "include"))
    if force or _stale(target, deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", target, src, "-L" + LIB, "-lptx355", "-lpthost",
                               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return target


def build_loopback(force: bool = False) -> str:
    """tests/loopback/librccl.so.1: the in-process RCCL stand-in the multi-rank GPU tests load (test infrastructure)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "loopback", "rccl_loopback.hip")
    target = os.path.join(root, "tests", "loopback", "librccl.so.1")
    if force or _stale(target, [src]):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-Wl,-soname,librccl.so.1", "-o", target, src])
    return target


def build_all(force: bool = False):
    out = build_device(force), build_host(force), build_harness(force)
    build_loopback(force)
    return out


if __name__ == "__main__":
    import sys
    print(build_all("--force" in sys.argv))

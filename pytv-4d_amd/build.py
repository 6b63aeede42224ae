#!/usr/bin/env python3
"""Build libpytv4d_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache: the .so sits next
to the Python package so that it travels with the repository snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tv_kernels.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "tv_device.h"), os.path.join(os.path.dirname(HERE), "include", "pytv4d.h")]
OUT = os.path.join(HERE, "pytv", "libpytv4d_hip.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-o", OUT, SRC]
    if verbose:
        print("[pytv build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)

#!/usr/bin/env python3
"""Build libpytv4d_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache: the .so sits next
to the Python package so that it travels with the repository snapshot to the GPU box.  The
translation units are compiled in parallel and linked with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# The product library: ONE generation of the one-pass sub-gradient kernel (tv_subgrad2.h).  The kernels that lost their A/B -- the
# round-1 kernel k_subgrad_one and the round-4 pair tile k_subgrad_pair -- were deleted in round 6 (git history: csrc/variants/).
UNITS = ["tv_kernels.hip", "tv_march_D.hip", "tv_march_DT.hip", "tv_fused.hip", "tv_fused_f64.hip", "tv_fused_admm.hip", "tv_fused_admm_f64.hip", "tv_fused_cpop.hip", "tv_fused_cpop_f64.hip", "tv_subgrad.hip", "tv_subgrad_norms.hip", "tv_sgstep.hip", "tv_dstream.hip", "tv_comm.hip", "tv_nstream.hip", "tv_small.hip"]
HEADERS = ["tv_device.h", "tv_stencil.h", "tv_host.h", "tv_march.h", "tv_fused.h", "tv_fused_launch.h", "tv_subgrad2.h", "tv_subgrad_host.h", "tv_dstream.h", "tv_nstream.h", "tv_site.h"]
DEPS = [os.path.join(CSRC, f) for f in UNITS + HEADERS] + [os.path.join(os.path.dirname(HERE), "include", "pytv4d.h")]
# TV_VARIANT=<name> (with TV_EXTRA_FLAGS): an experimental build next to the product library, loaded with PYTV4D_LIB=<path>
_VAR = os.environ.get("TV_VARIANT", "")
OUT = os.path.join(HERE, "pytv", "libpytv4d_hip%s.so" % (("_" + _VAR) if _VAR else ""))
OBJDIR = os.path.join(HERE, "build" + (("_" + _VAR) if _VAR else ""))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable", "-Wno-unused-function", "-I", CSRC]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def _compile(unit, verbose):
    obj = os.path.join(OBJDIR, os.path.basename(unit).replace(".hip", ".o"))
    cmd = [hipcc()] + FLAGS + os.environ.get("TV_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, unit), "-o", obj]
    if verbose:
        print("[pytv build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(lambda u: _compile(u, verbose), UNITS))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print("[pytv build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)

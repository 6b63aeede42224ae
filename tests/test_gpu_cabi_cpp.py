"""The C-ABI driven from C++ with plain hipMalloc'd buffers -- no Python, no PyTorch in the process: examples/cabi_demo.cpp
is compiled with hipcc against include/pytv4d.h + libpytv4d_hip.so and run (adjointness, the two CP paths, error status)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cabi_demo_builds_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    libdir = os.path.join(ROOT, "pytv-4d_amd", "pytv")
    exe = str(tmp_path / "cabi_demo")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "cabi_demo.cpp"), "-L" + libdir, "-lpytv4d_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi_demo: OK" in out.stdout
    assert "tv_small_cp (2 calls)" in out.stdout          # the persistent loop from C++: two calls on one workspace == five kernel-pair iterations

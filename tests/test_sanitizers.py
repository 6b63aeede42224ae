"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds (the GPU pool has no device ASan).
tools/sanitize.py builds (1) the C / OpenMP oracle with gcc -fsanitize=address,undefined and runs tests/test_oracle_c.py
against it, (2) the HOST side of every translation unit of the C-ABI with hipcc --cuda-host-only -fsanitize=... and drives
it through tools/abi_validation.py (argument validation, geometry rules, workspace sizing, option table -- everything an
entry point does before its first HIP call).  tools/abi_validation.py also runs on its own against the product library."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def test_abi_validation_against_the_product_library():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_validation.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "checks ok" in p.stdout


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs gcc and hipcc")
def test_sanitizer_builds_are_clean():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize.py")], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "oracle leg rc=0, C-ABI host leg rc=0" in p.stdout

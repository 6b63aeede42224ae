"""A seeded random walk over the dispatch of every entry point (``tools/stress_ops.py``): schemes, M = 1 .. 20, plane sizes on both
sides of the marching / one-sweep thresholds, z-chunk lengths, boolean masks / per-pixel weights / weight volumes, fp32 and fp64 --
D, D^T, TV + sub-gradient (with and without norms), Chambolle-Pock on both paths, ADMM in both CG forms and sub-gradient descent, each
against the CPU oracle.  The fixed-shape tests pin known corners; this walks the branches between them (it found three defects in
round 2 that the fixed shapes had missed).  A separate process: the walk sets library options as it goes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,n,last", [
    ("stress_ops.py", "160", "cases 160, mismatches 0"),           # every entry point against the oracle
    ("stress_subgrad.py", "80", "mismatches 0"),                   # one-pass sub-gradient against the two-pass kernels, twice (bitwise determinism)
    ("stress_small.py", "96", "mismatches 0"),                     # persistent small-volume loops (round 6) against the per-iteration kernels, twice
])
def test_random_walk_over_the_dispatch_matches_the_oracle(tool, n, last):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "pytv-4d_amd"), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), n], env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert last in r.stdout, tail

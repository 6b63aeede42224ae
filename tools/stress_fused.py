#!/usr/bin/env python3
"""Soak test of the one-sweep path (in-block LDS hand-off, lagged finalize): repeat many runs on awkward shapes and
compare against the two-kernel path run on the same input; any race would show as a mismatch on some repetition."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"
import torch, pytv
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "0")))
shapes = [(7, 8, 9, 320), (5, 3, 13, 260), (33, 8, 64, 512), (4, 4, 6, 1028), (9, 2, 31, 68),
          (6, 16, 10, 256), (5, 12, 7, 132), (4, 9, 5, 64), (3, 7, 6, 68), (17, 24, 8, 128)]        # M > 8: time windows
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for shape in shapes:
        for scheme in ("hybrid", "upwind", "downwind", "central"):
            x0 = torch.as_tensor((50 * rng.random(shape)).astype(np.float32)).cuda()
            a = pytv.solvers.ChambollePock(x0, 5.0, scheme=scheme, reg_time=0.7)
            b = pytv.solvers.ChambollePock(x0, 5.0, scheme=scheme, reg_time=0.7, fused=False)
            la, lb = a.run(6), b.run(6)
            ok = np.allclose(la, lb, rtol=2e-6) and torch.allclose(a.result(), b.result(), rtol=1e-5, atol=1e-3) \
                and torch.allclose(a.q, b.q, rtol=1e-5, atol=1e-4)
            if not ok:
                bad += 1
                print("MISMATCH rep %d %s %s: %s vs %s" % (rep, shape, scheme, la[-1], lb[-1]))
print("soak done: %d mismatches" % bad)
sys.exit(1 if bad else 0)

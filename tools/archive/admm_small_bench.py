#!/usr/bin/env python3
"""ADMM on a 512x512 image: eager launches vs hipGraph replay of blocks of outer iterations (solvers.ADMM.run(graph=...))."""
import os
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
rng = np.random.default_rng(0)
x0 = torch.as_tensor((rng.random((1, 1, 512, 512)) * 100).astype(np.float32)).cuda()
for g in (False, True):
    ad = pytv.solvers.ADMM(x0, 20.0, 0.1, n_cg=5)
    ad.run(12, graph=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 202
    ad.run(n, graph=g)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("ADMM 512x512, 5 CG steps, graph=%s: %.1f us per outer iteration (%.0f it/s)" % (g, dt * 1e6, 1 / dt))

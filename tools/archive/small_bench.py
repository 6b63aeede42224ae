#!/usr/bin/env python3
"""Launch-bound regime: README-sized 2-D problems (BASELINE config 0 is 512x512, 300 iterations)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from oracle import tv_oracle as orc
for n in (256, 512, 2048):
    rng = np.random.RandomState(0)
    noisy = (orc.phantom((1, 1, n, n), dtype=np.float64) + 100 * rng.rand(1, 1, n, n)).astype(np.float32)
    x0 = torch.as_tensor(noisy).cuda()
    for name, mk in (("CP one-sweep", lambda: pytv.solvers.ChambollePock(x0, 25.0)),
                     ("CP two-kernel", lambda: pytv.solvers.ChambollePock(x0, 25.0, fused=False)),
                     ("sub-gradient", lambda: pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3))):
        s = mk(); s.run(10); torch.cuda.synchronize()
        t0 = time.perf_counter(); loss = s.run(300); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%4d x %-4d %-14s 300 it in %.3f s  -> %.0f it/s (%.1f us/it)  loss %.6e" % (n, n, name, dt, 300 / dt, dt / 300 * 1e6, loss[-1]))
    if n <= 512:
        t0 = time.perf_counter(); orc.chambolle_pock(noisy.astype(np.float64), 300, 25.0); dt = time.perf_counter() - t0
        print("%4d x %-4d %-14s 300 it in %.3f s  -> %.0f it/s" % (n, n, "oracle CP (CPU)", dt, 300 / dt))

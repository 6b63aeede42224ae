#!/usr/bin/env python3
"""Round 6: the persistent small-volume loops (tv_small_cp / tv_small_subgrad_descent, csrc/tv_small.hip) against the ordinary small-volume
path (kernel pair / one-pass kernel per iteration, replayed from hipGraphs) on the reference's own shapes:
  (20,4,100,100)  README.md:76-79;   (1,1,512,512) / (1,1,256,256)  the README loops, 300 iterations (README.md:107-124, 141-157);
  (20,1,100,100)  pytv/tests.py:48.
usage: python tools/small_volume_bench.py [--iters 300] [NzxMxNyxNx ...]      -> one line per (shape, scheme, solver, path)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pytv

ITERS = 300
args = sys.argv[1:]
if "--iters" in args:
    i = args.index("--iters")
    ITERS = int(args[i + 1])
    del args[i:i + 2]
SCHEMES = ("hybrid", "upwind", "central")
if "--schemes" in args:
    i = args.index("--schemes")
    SCHEMES = tuple(args[i + 1].split(","))
    del args[i:i + 2]
SHAPES = [tuple(int(v) for v in a.split("x")) for a in args] or [(20, 4, 100, 100), (1, 1, 512, 512), (1, 1, 256, 256), (20, 1, 100, 100), (64, 4, 128, 128)]


def timed(make, n, reps=3):
    best, loss = 1e30, None
    for _ in range(reps):
        s = make()
        s.run(4)                      # warm-up of this instance (graph capture, workspace)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = s.run(n)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, loss


for shape in SHAPES:
    rng = np.random.default_rng(0)
    x0 = torch.as_tensor((100.0 * rng.random(shape)).astype(np.float32)).cuda()
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0 if shape[1] > 1 else 0.0)
    for scheme in SCHEMES:
        for name, mk in (("CP", lambda pers: pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, persistent=pers, **kw)),
                         ("SG", lambda pers: pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, persistent=pers, **kw))):
            t_old, l_old = timed(lambda: mk(False), ITERS)
            t_new, l_new = timed(lambda: mk(True), ITERS)
            rel = float(np.max(np.abs(l_new - l_old) / np.abs(l_old)))
            print("%-18s %-8s %s  %d iterations: ordinary path %8.3f ms (%6.2f us/it) | persistent %8.3f ms (%6.2f us/it) | x%.2f | max rel loss diff %.2e"
                  % ("x".join(map(str, shape)), scheme, name, ITERS, 1e3 * t_old, 1e6 * t_old / ITERS, 1e3 * t_new, 1e6 * t_new / ITERS, t_old / t_new, rel), flush=True)

#!/usr/bin/env python3
"""One-sweep Chambolle-Pock path against the dual + primal kernel pair on small / medium volumes (where does the default switch?).
usage: python tools/archive/fused_vs_pair.py [NzxMxNyxNx ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [
    (256, 1, 512, 512), (512, 1, 256, 256), (64, 1, 1024, 1024), (64, 4, 256, 256), (32, 8, 256, 256), (16, 4, 256, 256),
    (128, 2, 512, 512), (1, 1, 1024, 1024), (1, 1, 512, 512), (1, 8, 512, 512), (8, 1, 128, 128)]
dev = torch.device("cuda", 0)
for shape in shapes:
    x0 = synth_slab(shape, 0, shape[0], dev)
    res = {}
    for fused in (True, False):
        try:
            cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0 if shape[1] > 1 else 0.0, fused=fused)
        except ValueError as e:
            res[fused] = None
            continue
        n = 200 if x0.numel() < (1 << 26) else 40
        cp.run(24, record_loss=False, graph=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cp.run(n, record_loss=False, graph=False)
        torch.cuda.synchronize(); res[fused] = (time.perf_counter() - t0) / n * 1e6
        del cp
    plane_kb = shape[1] * shape[2] * shape[3] * 4 // 1024
    f, p = res[True], res[False]
    print("%-20s plane %6d KiB, %7.1f Mvox: one-sweep %9s us  pair %9.1f us  -> %s" % (
        "x".join(map(str, shape)), plane_kb, np.prod(shape) / 1e6, ("%.1f" % f) if f else "n/a", p,
        "n/a" if not f else ("one-sweep %.2fx" % (p / f))), flush=True)
    del x0
    torch.cuda.empty_cache()

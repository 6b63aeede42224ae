#!/usr/bin/env python3
"""Chambolle-Pock with a data-fidelity operator (solvers.ChambollePockOperator, A = a diagonal operator written with torch):
the TV part as one sweep (tv_cpop_fused + tv_cpop_fixup) against the kernel pair tv_cp_dual + tv_DT_axpy2.
usage: python tools/archive/cpop_bench.py [NzxMxNyxNx] [scheme ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (64, 8, 1024, 1024)
schemes = sys.argv[2:] or ["hybrid", "upwind", "central"]
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
a = 0.2 + 0.8 * torch.rand(shape, device=dev)
b = a * x0 + 0.5
bufA, bufAT = torch.empty_like(x0), torch.empty_like(x0)
V = x0.numel()
for scheme in schemes:
    for fused in (True, False):
        cp = pytv.solvers.ChambollePockOperator(lambda v: torch.mul(a, v, out=bufA), lambda v: torch.mul(a, v, out=bufAT), b, x0, 25.0,
                                                 scheme=scheme, reg_time=1.0, fused=fused)
        hist = torch.zeros((16, 2), dtype=torch.float64, device=dev)
        for i in range(3):
            cp.step(hist[i])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            cp.step(hist[3 + i])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        nd = cp.geo.nd
        # TV part: fused 2 Nd + 4 (q r/w, x, A^T p read, x written, + fix-up), pair 3 Nd + 4; the diagonal A, A^T, p update and residual
        # of this bench add 3 + 3 + 3 + 3 words
        words = (2 * nd + 4 if fused else 3 * nd + 4) + 12
        print("%-8s %-9s %.2f ms/it  (%d words/voxel incl. the torch operator: %.0f GB/s)" % (scheme, "one-sweep" if fused else "pair", dt * 1e3, words,
                                                                                             words * 4.0 * V / dt / 1e9))
        del cp

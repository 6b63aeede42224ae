#!/usr/bin/env python3
"""ADMM outer iterations/s (BASELINE config 4 flavour: M = 16, all four discretisations) on one GPU.
usage: python tools/admm_bench.py [NzxMxNyxNx] [n_cg]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (32, 16, 1024, 1024)
n_cg = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
V = x0.numel()
print("ADMM on %s (V = %.0f Mvox), %d CG steps per outer iteration, rho = 0.05, lambda = 25" % (shape, V / 1e6, n_cg))
for scheme in ("upwind", "downwind", "central", "hybrid"):
    ad = pytv.solvers.ADMM(x0, 25.0, 0.05, n_cg=n_cg, scheme=scheme, reg_time=1.0)
    ad.run(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = ad.run(4)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    nd = ad.geo.nd
    # words/voxel per outer iteration: rhs (2Nd+2) + (n_cg+1) normal ops (2) + n_cg (cg1: 6, cg2: 3) + setup (sub 3, copy 2, dot 2)
    #                                  + zu (1+3Nd) + fidelity (sub 3 + dot 2)
    words = (2 * nd + 2) + (n_cg + 1) * 2 + n_cg * 9 + 7 + (1 + 3 * nd) + 5
    print("%-9s Nd=%d  %.2f ms/outer  %.1f it/s  loss %.6e -> %.6e  | algorithmic %.0f words/voxel -> %.0f GB/s" % (
        scheme, nd, dt * 1e3, 1 / dt, loss[0], loss[-1], words, words * 4.0 * V / dt / 1e9))
    del ad
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Sub-gradient descent loop (README.md:118-124) on the GPU: iterations/s of the one-pass kernel (TV + G + step in one
sweep) against the two-pass tv_subgrad + tv_subgrad_step.  usage: python tools/sg_bench.py [NzxMxNyxNx] [scheme ...]
SG_TUNE=0 / 1 in the environment: placement tuner of the solver off / on (default: the solver's own rule)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (256, 8, 1024, 1024)
schemes = sys.argv[2:] or ["hybrid", "upwind", "downwind", "central"]
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
V = float(x0.numel())
print("shape %s; words = fp32 words per voxel the iteration must move (x, x0 read, x written = 3)" % (shape,))
for scheme in schemes:
    for one_pass in (True, False):
        tune = os.environ.get("SG_TUNE")
        sg = pytv.solvers.SubgradientDescent(x0, 25.0, 0.01, scheme=scheme, reg_time=1.0, one_pass=one_pass, tune_placement=None if tune is None else bool(int(tune)))
        hist = torch.zeros((16, sg.SLOTS), dtype=torch.float64, device=x0.device)
        for i in range(3):
            sg.step(hist[i])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            sg.step(hist[3 + i])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("%-9s %-8s %8.2f ms/it  %7.2f it/s  %6.0f GB/s algorithmic (frac of 8 TB/s %.3f)" % (
            scheme, "one-pass" if one_pass else "two-pass", dt * 1e3, 1 / dt, 12 * V / dt / 1e9, 12 * V / dt / 8e12))
        if sg.placement:
            print("          placement tuner: %s" % ({k: sg.placement.get(k) for k in ("step_ms", "chosen", "first_pair_ms", "chosen_ms", "x0_round_trip_ms", "seconds", "error") if k in sg.placement},))
        del sg
        torch.cuda.empty_cache()

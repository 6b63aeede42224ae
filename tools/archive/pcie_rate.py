#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-in / numpy-out boundary (what the reference's README loops pay on
every call) next to the device-resident rate, for DESIGN.md.  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
import torch
import pytv
shape = (32, 8, 512, 512)
x = np.random.default_rng(0).random(shape, dtype=np.float32)
V = x.size
ops = pytv.tv_operators_GPU
ops.D_hybrid(x[:2], reg_time=1.0)
t0 = time.perf_counter(); d = ops.D_hybrid(x, reg_time=1.0); t1 = time.perf_counter()
print("numpy in/out   D_hybrid %s: %.3f s  -> %.1f Mvox/s, %.2f GB/s over the boundary (x H2D + D D2H = %.2f GB)" % (shape, t1 - t0, V / (t1 - t0) / 1e6, (x.nbytes + d.nbytes) / (t1 - t0) / 1e9, (x.nbytes + d.nbytes) / 1e9))
t0 = time.perf_counter(); o = ops.D_T_hybrid(d, reg_time=1.0); t1 = time.perf_counter()
print("numpy in/out   D_T_hybrid: %.3f s -> %.1f Mvox/s" % (t1 - t0, V / (t1 - t0) / 1e6))
t0 = time.perf_counter(); tv, G = pytv.tv_GPU.tv_hybrid(x, reg_time=1.0); t1 = time.perf_counter()
print("numpy in/out   tv_hybrid : %.3f s -> %.1f Mvox/s" % (t1 - t0, V / (t1 - t0) / 1e6))
xt = torch.as_tensor(x).cuda(); torch.cuda.synchronize()
for name, f in (("D_hybrid", lambda: ops.D_hybrid(xt, reg_time=1.0)), ("tv_hybrid", lambda: pytv.tv_GPU.tv_subgradient_device(xt, "hybrid", reg_time=1.0))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): r = f()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("device resident %s: %.2f ms -> %.0f Mvox/s" % (name, (t1 - t0) / 5 * 1e3, V / ((t1 - t0) / 5) / 1e6))
dt = ops.D_hybrid(xt, reg_time=1.0)
f = lambda: ops.D_T_hybrid(dt, reg_time=1.0)
f(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("device resident D_T_hybrid: %.2f ms -> %.0f Mvox/s" % ((t1 - t0) / 5 * 1e3, V / ((t1 - t0) / 5) / 1e6))

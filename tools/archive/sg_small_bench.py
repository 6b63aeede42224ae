#!/usr/bin/env python3
"""TV + sub-gradient: the one-pass kernel against the two-pass forms on small and mid-size volumes.
usage: python tools/archive/sg_small_bench.py [NzxMxNyxNx ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
def bench(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(1,1,512,512), (1,1,2048,2048), (256,1,512,512), (128,8,512,512), (64,1,1024,1024), (16,4,256,256), (32,2,1024,1024)]
for shape in SHAPES:
    x = torch.rand(shape, device="cuda") * 100
    for scheme in ("hybrid", "upwind"):
        kw = dict(reg_time=1.0 if shape[1] > 1 else 0.0)
        t1 = bench(lambda: pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=False, one_pass=True, **kw))
        pytv._native.set_option("TV_MARCH_MIN_PLANE_KB", 0)
        t2m = bench(lambda: pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=True, one_pass=False, **kw))
        pytv._native.set_option("TV_MARCH_MIN_PLANE_KB", 1000000)
        t2g = bench(lambda: pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=True, one_pass=False, **kw))
        pytv._native.set_option("TV_MARCH_MIN_PLANE_KB", None)
        print("%-20s %-8s one-pass %.3f ms | two-pass marching %.3f | two-pass generic %.3f" % (shape, scheme, t1, t2m, t2g))

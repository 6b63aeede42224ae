"""What does ONE drop-in call cost on a small image (numpy in / numpy out, the reference's README loop)?  usage: python tools/archive/r6_shim_overhead.py"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
rng = np.random.default_rng(0)
for shape, dt in (((1, 1, 256, 256), np.float64), ((1, 1, 512, 512), np.float32), ((20, 4, 100, 100), np.float32)):
    x = (rng.random(shape) * 100).astype(dt)
    kw = dict(reg_time=1.0) if shape[1] > 1 else {}
    for _ in range(5):
        pytv.tv_GPU.tv_hybrid(x, **kw)
    t0 = time.perf_counter()
    for _ in range(200):
        tv, G = pytv.tv_GPU.tv_hybrid(x, **kw)
    t1 = time.perf_counter()
    d = pytv.tv_operators_GPU.D_hybrid(x, **kw)
    for _ in range(3):
        pytv.tv_operators_GPU.D_hybrid(x, **kw); pytv.tv_operators_GPU.D_T_hybrid(d, **kw)
    t2 = time.perf_counter()
    for _ in range(100):
        d = pytv.tv_operators_GPU.D_hybrid(x, **kw)
    t3 = time.perf_counter()
    for _ in range(100):
        pytv.tv_operators_GPU.D_T_hybrid(d, **kw)
    t4 = time.perf_counter()
    print("%-16s %-8s tv_hybrid %7.1f us/call | D_hybrid %7.1f | D_T_hybrid %7.1f" % ("x".join(map(str, shape)), np.dtype(dt).name, 1e6 * (t1 - t0) / 200, 1e6 * (t3 - t2) / 100, 1e6 * (t4 - t3) / 100), flush=True)
x = (rng.random((1, 1, 256, 256)) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    pytv.tv_GPU.tv_hybrid(x)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])

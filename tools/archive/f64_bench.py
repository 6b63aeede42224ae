import sys, os, time
sys.path.insert(0, "pytv-4d_amd"); sys.path.insert(0, ".")
import torch, pytv
from bench import synth_slab
shape = (64, 8, 1024, 1024)
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
for dt in (torch.float32, torch.float64):
    xx = x0.to(dt)
    for fused in (True, False):
        cp = pytv.solvers.ChambollePock(xx, 25.0, reg_time=1.0, fused=fused)
        for _ in range(2): cp.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): cp.step()
        torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / 6
        words = (5 + 16) if cp.fused else (6 + 24)
        print(dt, "fused" if cp.fused else "two-kernel", "%.2f ms/it" % (dt_s * 1e3), "%.0f GB/s algorithmic" % (words * xx.element_size() * xx.numel() / dt_s / 1e9))
        del cp; torch.cuda.empty_cache()
    tv, G, _ = pytv.tv_GPU.tv_subgradient_device(xx, "hybrid", reg_time=1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tv, G, _ = pytv.tv_GPU.tv_subgradient_device(xx, "hybrid", reg_time=1.0)
    torch.cuda.synchronize(); print(dt, "tv_subgrad two-pass %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    del xx

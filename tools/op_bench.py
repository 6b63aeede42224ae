#!/usr/bin/env python3
"""Achieved bandwidth of every C-ABI operator (device resident, fp32) against its algorithmic bytes.
usage: python tools/op_bench.py [NzxMxNyxNx] [scheme ...]      (DTYPE=f64 in the environment: double precision)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch
import pytv  # noqa: F401  (loads the library)
from pytv import _native as nv
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (64, 8, 1024, 1024)
schemes = sys.argv[2:] or ["hybrid", "upwind", "downwind", "central"]
lib = nv.lib()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(shape, device=dev, generator=gen) * 100
if os.environ.get("DTYPE", "f32") == "f64":
    x = x.double()
V = x.numel()
WB = x.element_size()


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


print("shape %s  V = %.0f Mvox  dtype %s; GB/s = algorithmic bytes / time; frac of 8000 GB/s" % (shape, V / 1e6, x.dtype))
print("%-9s %-22s %8s %9s %7s  %s" % ("scheme", "op", "ms", "GB/s", "frac", "algorithmic words/voxel"))
for scheme in schemes:
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    if os.environ.get("WEIGHT", "") == "vol":       # per-voxel weight on the time channels: sqrt(W) resident on the device
        kw["weight_dev"] = (torch.sqrt(torch.rand(shape, device=dev, generator=gen) * 2.0).to(x.dtype), None, None)
    g = nv.Geometry(shape, scheme, x.dtype, dev, **kw)
    nd, st, ws = g.nd, nv.current_stream(dev), g.workspace()
    d = torch.empty(g.grad_shape, device=dev, dtype=x.dtype)
    o = torch.empty_like(x); o2 = torch.empty_like(x)
    ne = torch.empty((shape[0] + 2,) + shape[1:], device=dev, dtype=x.dtype)
    sc = g.scalar()
    u = torch.zeros(g.grad_shape, device=dev, dtype=x.dtype)
    ops = [
        ("tv_D", 1 + nd, lambda: nv.check(lib.tv_D(g.ref, nv.ptr(x), None, None, nv.ptr(d), st))),
        ("tv_DT", nd + 1, lambda: nv.check(lib.tv_DT(g.ref, nv.ptr(d), None, None, nv.ptr(o), st))),
        ("tv_l21", nd, lambda: nv.check(lib.tv_l21(g.ref, nv.ptr(d), nd, None, nv.ptr(sc), nv.ptr(ws), st))),
        ("tv_subgrad", 2, lambda: nv.check(lib.tv_subgrad(g.ref, nv.ptr(x), None, None, nv.ptr(o), nv.ptr(ne), nv.ptr(sc), nv.ptr(ws), st))),
        ("tv_subgrad_fused", 2, (lambda: nv.check(lib.tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(o), nv.ptr(sc), nv.ptr(ws), st)))
         if lib.tv_subgrad_fused_supported(g.ref) else None),
        ("tv_subgrad_fused_norms", 3, (lambda: nv.check(lib.tv_subgrad_fused_norms(g.ref, nv.ptr(x), None, None, nv.ptr(o), nv.ptr(o2), nv.ptr(sc), nv.ptr(ws), st)))
         if lib.tv_subgrad_fused_supported(g.ref) else None),
        ("tv_normal_op", 2, lambda: nv.check(lib.tv_normal_op(g.ref, nv.ptr(x), None, None, 0.1, nv.ptr(o), nv.ptr(sc), nv.ptr(ws), st))),
        ("tv_admm_zu", 1 + 3 * nd, lambda: nv.check(lib.tv_admm_zu(g.ref, nv.ptr(x), None, None, nv.ptr(d), nv.ptr(u), 1.0, nv.ptr(sc), nv.ptr(ws), st))),
        ("tv_DT_axpy", 2 * nd + 2, lambda: nv.check(lib.tv_DT_axpy(g.ref, nv.ptr(d), nv.ptr(u), None, None, nv.ptr(x), 0.1, nv.ptr(o), st))),
        ("tv_cp_dual", 1 + 2 * nd, lambda: nv.check(lib.tv_cp_dual(g.ref, nv.ptr(x), None, None, nv.ptr(d), 0.5, 25.0, nv.ptr(sc), nv.ptr(ws), st))),
        ("tv_cp_primal", nd + 5, lambda: nv.check(lib.tv_cp_primal(g.ref, nv.ptr(d), None, None, nv.ptr(o), nv.ptr(x), nv.ptr(o2), 0.05, 1.0, nv.ptr(sc), nv.ptr(ws), st))),
    ]
    wvol = 1 if "weight_dev" in kw else 0          # one more word per voxel wherever the time channels are formed

    def cp_sweep():
        nv.check(lib.tv_cp_fused(g.ref, nv.ptr(x), None, None, nv.ptr(d), nv.ptr(x), nv.ptr(o2), nv.ptr(o), 0.5, 25.0, 1.0 / 17.0, 1.0, 0, -1,
                                 nv.ptr(sc), nv.ptr(sc), nv.ptr(ws), st))
        nv.check(lib.tv_cp_fixup(g.ref, nv.ptr(d), None, None, nv.ptr(o), nv.ptr(x), 1.0 / 17.0, 0, -1, nv.ptr(sc), nv.ptr(ws), st))

    def admm_sweep():          # z / u update + residual of the next x-solve (u in place, t' stored sparsely)
        nv.check(lib.tv_admm_fused(g.ref, nv.ptr(x), None, None, nv.ptr(u), nv.ptr(d), nv.ptr(o2), nv.ptr(o), 1.0, 0.05, 0, 0, -1,
                                   nv.ptr(sc), nv.ptr(sc), nv.ptr(ws), st))
        nv.check(lib.tv_admm_fixup(g.ref, nv.ptr(d), None, None, nv.ptr(o), 0.05, 0, -1, nv.ptr(sc), nv.ptr(ws), st))

    def cpop_sweep():          # operator-slot CP: dual update + x - tau A^T p - tau D^T q
        nv.check(lib.tv_cpop_fused(g.ref, nv.ptr(x), None, None, nv.ptr(d), nv.ptr(o2), nv.ptr(o), 0.5, 25.0, 1.0 / 17.0, 0, -1, nv.ptr(sc),
                                   nv.ptr(ws), st))
        nv.check(lib.tv_cpop_fixup(g.ref, nv.ptr(d), None, None, nv.ptr(o), 1.0 / 17.0, 0, -1, nv.ptr(ws), st))

    dots2 = torch.zeros(2, dtype=torch.float64, device=dev)
    ops.append(("tv_cheb_step", 4, lambda: nv.check(lib.tv_cheb_step(g.ref, nv.ptr(x), None, None, 0.05, nv.ptr(o2), nv.ptr(ne[1:1 + shape[0]]), 0.0, None, None,
                                                                     0.8, 0.3, nv.ptr(o), dots2.data_ptr(), nv.ptr(ws), st))))
    if lib.tv_cp_fused_supported(g.ref):
        ops.append(("cp_sweep+fixup", 5 + 2 * nd + wvol, cp_sweep))
        ops.append(("admm_sweep+fixup", 3 + 2 * nd + wvol, admm_sweep))
        ops.append(("cpop_sweep+fixup", 3 + 2 * nd + wvol, cpop_sweep))
    if wvol:
        ops = [(n, wd + (1 if n in ("tv_D", "tv_subgrad_fused", "tv_subgrad_fused_norms", "tv_cp_dual", "tv_subgrad", "tv_admm_zu") else 0), f) for n, wd, f in ops]
    only = [o for o in os.environ.get("OPS", "").split(",") if o]
    for name, words, f in ops:
        if (only and name not in only) or f is None:
            continue
        t = timeit(f)
        gbs = words * float(WB) * V / t / 1e9
        print("%-9s %-22s %8.3f %9.0f %7.3f  %d" % (scheme, name, t * 1e3, gbs, gbs / 8000.0, words))
    del d, u, o, o2, ne
    torch.cuda.empty_cache()

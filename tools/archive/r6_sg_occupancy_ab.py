import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch
from pytv import _native as nv
shape = (64, 8, 1024, 1024)
x = torch.rand(shape, device="cuda") * 100
lib = nv.lib()
for scheme in ("hybrid", "upwind", "downwind", "central"):
    geo = nv.Geometry(shape, scheme, x.dtype, x.device, 1.0, 1.0, False, 0)
    G = torch.empty_like(x); tv = geo.scalar(); ws = geo.workspace(); st = nv.current_stream(x.device)
    for _ in range(3):
        nv.check(lib.tv_subgrad_fused(geo.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(ws), st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        nv.check(lib.tv_subgrad_fused(geo.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(ws), st))
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("%-9s tv_subgrad_fused %.3f ms  (%.3f of 8 TB/s at 8 B / voxel)  TV %.6e" % (scheme, ms, 8.0 * x.numel() / (ms * 1e-3) / 8e12, float(tv)))

#!/usr/bin/env python3
"""Does the ADMM outer iteration depend on where its arrays landed (as the Chambolle-Pock sweep does)?  Construct the
solver several times in one process -- the previous instance is kept alive while the next one allocates, so that the allocations differ --
and time the same iterations on each.  usage: python tools/admm_placement_probe.py [NzxMxNyxNx=32x16x1024x1024] [scheme=hybrid] [n=5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (32, 16, 1024, 1024)
scheme = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pre = float(os.environ.get("PREALLOC_GIB", "0"))
if pre > 0:        # does memory that this process has already had once answer faster than first-time allocations?
    t = torch.empty(int(pre * 2 ** 30), dtype=torch.uint8, device="cuda")
    if os.environ.get("PRETOUCH", "1") == "1":
        t.zero_()
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
    print("pre-allocated and freed %.0f GiB (touched: %s)" % (pre, os.environ.get("PRETOUCH", "1")), flush=True)
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
prev = None
for k in range(n):
    ad = pytv.solvers.ADMM(x0, 25.0, 0.05, n_cg=5, scheme=scheme, reg_time=1.0, tune_placement=(os.environ.get("ADMM_TUNE", "0") == "1"))
    if ad.placement: print("   tuner:", ad.placement)
    ad.run(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ad.run(8)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print("instance %d: %.3f ms per outer iteration (u at %#x, t' at %#x, x at %#x)" % (k, dt * 1e3, ad.u.data_ptr(), ad._zt.data_ptr(), ad.x.data_ptr()), flush=True)
    prev = ad if k % 2 == 0 else None          # keep every other instance alive: the next one cannot reuse its blocks
    del ad
    torch.cuda.empty_cache()

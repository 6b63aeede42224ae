#!/usr/bin/env python3
"""BASELINE config 3 on ONE GPU: (512, 8, 1024, 1024) fp32 = 2^32 voxels, a 2^35-element dual variable (192 GiB resident).
Three Chambolle-Pock iterations through the one-sweep path and through the two-kernel path must give the same loss:
two independent sets of kernels agreeing beyond 2^32 elements is the 64-bit indexing check at full size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (512, 8, 1024, 1024)
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
losses = {}
for name, fused in (("one-sweep", None), ("two-kernel", False)):
    cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, fused=fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses[name] = cp.run(3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    tail = cp.result()[-1, -1, -1, -8:].cpu().numpy()          # the very last elements of the volume
    print("%-10s %.1f ms/it  loss %s  x[-8:] %s" % (name, dt * 1e3, np.array2string(losses[name], precision=9), np.array2string(tail, precision=4)))
    del cp
    torch.cuda.empty_cache()
a, b = losses["one-sweep"], losses["two-kernel"]
ok = np.all(np.isfinite(a)) and np.allclose(a, b, rtol=1e-6) and a[-1] < a[0]
print("agree" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Does the one-sweep kernel's time depend on how its arrays are ALIGNED relative to each other?  The five arrays (x, x_alt, p, x0,
q) are re-sliced out of padded buffers at byte offsets k * S (k = 0..4) inside ONE process, so the physical pages stay the same and
only the relative alignment changes.
usage: python tools/offset_probe.py [S_bytes ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
dev = torch.device("cuda", 0)
shape = (256, 8, 1024, 1024)
PAD = 64 << 20                       # bytes of slack per array
x0 = synth_slab(shape, 0, shape[0], dev)
cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0)
n = x0.numel()
bufs = {k: torch.zeros(n + PAD // 4, dtype=torch.float32, device=dev) for k in ("x", "x_alt", "p", "x0")}
qn = cp.q.numel()
qbuf = torch.zeros(qn + PAD // 4, dtype=torch.float32, device=dev)
src = {"x": cp.x.clone(), "x0": cp.x0.clone()}
del cp.q
torch.cuda.empty_cache()
steps = [int(v) for v in sys.argv[1:]] or [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, (2 << 20) + 65536 + 4096]
for rep in range(2):
    for S in steps:
        offs = {"x": 0, "x_alt": S, "p": 2 * S, "x0": 3 * S, "q": 4 * S}
        for k in ("x", "x_alt", "p", "x0"):
            o = offs[k] // 4
            setattr(cp, k, bufs[k][o:o + n].view(shape))
        o = offs["q"] // 4
        cp.q = qbuf[o:o + qn].view(cp.geo.grad_shape)
        cp.x.copy_(src["x"]); cp.x0.copy_(src["x0"]); cp.p.zero_(); cp.q.zero_()
        for _ in range(2):
            cp.step()
        cp.timing = []
        for _ in range(8):
            cp.step()
        torch.cuda.synchronize()
        k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]
        cp.timing = None
        print("S = %8d B: sweep ms even/odd iterations %.2f / %.2f   (base pointers mod 2 MiB: %s)" % (
            S, np.mean(k1[0::2]), np.mean(k1[1::2]),
            " ".join("%s=%d" % (k, getattr(cp, k).data_ptr() % (2 << 20)) for k in ("x", "x_alt", "p", "x0", "q"))), flush=True)

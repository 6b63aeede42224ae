#!/usr/bin/env python3
"""Where does an iteration of the persistent Chambolle-Pock kernel spend its time?  Needs a -DTV_SMALL_PROFILE variant build:
    TV_VARIANT=smallprof TV_EXTRA_FLAGS=-DTV_SMALL_PROFILE python3 pytv-4d_amd/build.py
    PYTV4D_LIB=pytv-4d_amd/pytv/libpytv4d_hip_smallprof.so python3 tools/small_volume_profile.py 20x4x100x100 hybrid
Thread 0 of every block records the 100 MHz wall clock at five points of every iteration; this prints, over blocks and iterations,
the mean / max of: dual phase, wait after it, primal phase, wait after it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pytv
from pytv import _native as nv

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20x4x100x100").split("x"))
scheme = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
rng = np.random.default_rng(0)
x0 = torch.as_tensor((100.0 * rng.random(shape)).astype(np.float32)).cuda()
cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_z_over_reg=1.0, reg_time=1.0 if shape[1] > 1 else 0.0, persistent=True)
K = cp.SMALL_BLOCK
cp.run(K)
cp.run(K)
torch.cuda.synchronize()
ws = cp._small_ws().cpu().numpy()
flags_doubles = (8192 + 1) * 32 * 4 // 8
body = ws[flags_doubles:]
# nblocks: the partial rows are contiguous; the marks follow them.  Find nblocks from the first clock value (> 1e6) position
pos = int(np.argmax(body > 1e9))
nblocks = pos // (2 * K)
marks = body[pos:pos + K * nblocks * 5].reshape(K, nblocks, 5)
d = np.diff(marks, axis=2) * 1e-2            # us (100 MHz)
names = ("dual phase", "wait for neighbours", "primal phase", "wait for neighbours")
print("%s %s: %d blocks, %d iterations per launch" % ("x".join(map(str, shape)), scheme, nblocks, K))
for k, n in enumerate(names):
    v = d[1:-1, :, k]
    print("  %-22s mean %6.2f us | median %6.2f | max over blocks (mean over iterations) %6.2f | min %6.2f" % (n, v.mean(), np.median(v), v.max(axis=1).mean(), v.min(axis=1).mean()))
it_time = (marks[1:, :, 0] - marks[:-1, :, 0]) * 1e-2
print("  iteration (start to start): mean %6.2f us" % it_time.mean())
# per-block view: are the slow blocks always the same ones (structural) or is it noise?
pb = d[1:-1].mean(axis=0)             # (nblocks, 4): mean over iterations
for k, n in enumerate(names):
    v = pb[:, k]
    qs = np.percentile(v, [0, 10, 50, 90, 100])
    print("  per-block mean of %-22s min %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % (n, *qs))
noise = d[1:-1].std(axis=0).mean(axis=0)
print("  std over iterations (mean over blocks):", " ".join("%.2f" % v for v in noise))
ph = pb[:, 0]
order = np.argsort(-ph)
print("  slowest 16 blocks (logical id: mean dual phase us):", " ".join("%d:%.2f" % (i, ph[i]) for i in order[:16]))
print("  fastest 8 blocks:", " ".join("%d:%.2f" % (i, ph[i]) for i in order[-8:]))
# absolute skew: when does each block start an iteration relative to the earliest block?
st = marks[1:-1, :, 0] - marks[1:-1, :, 0].min(axis=1, keepdims=True)
print("  start skew across blocks within an iteration: mean %.2f us, max %.2f us" % (st.mean() * 1e-2, st.max(axis=1).mean() * 1e-2))

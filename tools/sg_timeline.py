#!/usr/bin/env python3
"""Per-plane timeline of the one-pass sub-gradient kernel (round-5 verdict item 2: where do the cycles of k_subgrad_col go that its issue
floor does not explain?).  Needs a variant build with the marks compiled in:
    TV_VARIANT=sgtl TV_EXTRA_FLAGS=-DTV_SG2_TIMELINE python3 pytv-4d_amd/build.py
    PYTV4D_LIB=pytv-4d_amd/pytv/libpytv4d_hip_sgtl.so python3 tools/sg_timeline.py [NzxMxNyxNx] [scheme]
Lane 0 of every wave of the first 256 blocks records the shader clock (s_memtime) at the top of every plane step, before the LDS barrier
that ends the step and after the barrier.  Printed: per step and wave -- cycles from the top to the barrier (the wave's own work: issue +
every s_waitcnt it sits in), cycles in the barrier (waiting for the slowest wave of the block), and how the waves of a block differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pytv
from pytv import _native as nv

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x8x1024x1024").split("x"))
scheme = sys.argv[2] if len(sys.argv) > 2 else "hybrid"
x = torch.rand(shape, device="cuda") * 100
geo = nv.Geometry(shape, scheme, x.dtype, x.device, 1.0, 1.0, False, 0)
G = torch.empty_like(x)
tv = geo.scalar()
ws = geo.workspace()
lib = nv.lib()
st = nv.current_stream(x.device)
for _ in range(3):
    ws.zero_()
    nv.check(lib.tv_subgrad_fused(geo.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(ws), st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
nv.check(lib.tv_subgrad_fused(geo.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(ws), st))
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b)
third = ws.numel() // 3
raw = ws[third:2 * third].cpu().numpy().view(np.uint64)
NWAVES, STEPS = 8, 40
m = raw[:256 * NWAVES * STEPS * 4].reshape(256, NWAVES, STEPS, 4).astype(np.float64)
used = (m[..., 0] > 0) & (m[..., 2] > 0)
nsteps = int(used[0, 0].sum())
m = m[:, :, :nsteps]
work = m[..., 1] - m[..., 0]                      # top of the step -> barrier
barr = m[..., 2] - m[..., 1]                      # inside s_waitcnt lgkmcnt(0) + s_barrier
step = np.diff(m[..., 0], axis=2)                 # top -> next top
wall = m[..., 3]
ratio = (m[:, :, -1, 0] - m[:, :, 0, 0]) / np.maximum(1.0, wall[:, :, -1] - wall[:, :, 0])      # s_memtime ticks per 10 ns
mhz = 100.0 * float(np.median(ratio))
print("%s %s: tv_subgrad_fused %.3f ms (with the marks); %d plane steps per chunk; s_memtime runs at %.0f MHz during the kernel (against the 100 MHz wall clock)" % ("x".join(map(str, shape)), scheme, ms, nsteps, mhz))
print("  i.e. a plane step of %.0f cycles takes %.2f us; a block's plane loop %.1f us" % (step[:, :, 1:].mean(), step[:, :, 1:].mean() / mhz, (m[:, :, -1, 2].max(axis=1) - m[:, :, 0, 0].min(axis=1)).mean() / mhz))
dev = torch.cuda.get_device_properties(0)
print("  per step and wave: own work %7.0f cycles (median %7.0f, p10 %7.0f, p90 %7.0f)" % (work[:, :, 1:].mean(), np.median(work[:, :, 1:]), np.percentile(work[:, :, 1:], 10), np.percentile(work[:, :, 1:], 90)))
print("                     in the barrier %7.0f cycles (median %7.0f, p90 %7.0f) = %.1f %% of a step" % (barr[:, :, 1:].mean(), np.median(barr[:, :, 1:]), np.percentile(barr[:, :, 1:], 90), 100 * barr[:, :, 1:].mean() / step[:, :, 1:].mean()))
print("                     step (top to top) %7.0f cycles" % step[:, :, 1:].mean())
spread = work[:, :, 1:].max(axis=1) - work[:, :, 1:].min(axis=1)
print("  slowest minus fastest wave of a block, per step: mean %7.0f cycles (the barrier makes every wave wait for the slowest)" % spread.mean())
per_wave = work[:, :, 1:].mean(axis=(0, 2))
print("  own work by wave index (0..3 first wave column, 4..7 second): " + " ".join("%.0f" % v for v in per_wave))
first = work[:, :, 0].mean()
print("  first step of a chunk (prologue loads outstanding): own work %7.0f cycles" % first)
blk = (m[:, :, -1, 2].max(axis=1) - m[:, :, 0, 0].min(axis=1))
print("  a block's plane loop: %7.0f cycles (mean over 256 blocks; min %7.0f, max %7.0f)" % (blk.mean(), blk.min(), blk.max()))

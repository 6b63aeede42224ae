#!/usr/bin/env python3
"""Round 4: q ping-pong on the REAL one-sweep kernel.  tools/bwtest4 says "read one q array, write another" is 9 % faster than the
in-place read-modify-write for the sweep's memory shape, but with q and q_alt as two separate allocations the sweep alternates
between ~32 ms (one direction) and ~35 ms (the other).  Here both arrays live in ONE allocation at a chosen distance delta, and the
sweep is timed per direction for several deltas (and in place, in the same pool).
usage: python tools/pp_probe.py [--shape 256x8x1024x1024] [--deltas 0,4096,...] (bytes added to the dense back-to-back distance)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x8x1024x1024")
ap.add_argument("--deltas", default="0,4096,69888,1048576,2097152,3145728,34607360,1073741824,1077940480")
ap.add_argument("--steps", type=int, default=8)
args = ap.parse_args()
shape = tuple(int(v) for v in args.shape.split("x"))
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, fused=True, q_pingpong=False)
qshape = tuple(cp.q.shape)
qn = cp.q.numel()
del cp.q
torch.cuda.empty_cache()
deltas = [int(v) for v in args.deltas.split(",")]
pool = torch.zeros(2 * qn + max(deltas) // 4 + 1024, dtype=torch.float32, device=dev)
print("# pool %.1f GiB at %s; x %s x_alt %s p %s x0 %s" % (pool.numel() * 4 / 2 ** 30, hex(pool.data_ptr()), hex(cp.x.data_ptr()), hex(cp.x_alt.data_ptr()),
                                                     hex(cp.p.data_ptr()), hex(cp.x0.data_ptr())), flush=True)


def run(tag):
    for _ in range(4):
        cp.step()
    cp.timing = []
    torch.cuda.synchronize()
    for _ in range(args.steps):
        cp.step()
    torch.cuda.synchronize()
    k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]
    k2 = [e[1].elapsed_time(e[2]) for e in cp.timing]
    cp.timing = None
    print(json.dumps({"case": tag, "sweep_even": round(float(np.mean(k1[0::2])), 3), "sweep_odd": round(float(np.mean(k1[1::2])), 3),
                      "sweep_mean": round(float(np.mean(k1)), 3), "fixup": round(float(np.mean(k2)), 3)}), flush=True)


for rep in range(2):
    cp.q, cp.q_alt = pool[:qn].view(qshape), None
    run("in place (q at the start of the pool)")
    for d in deltas:
        off = qn + d // 4
        cp.q, cp.q_alt = pool[:qn].view(qshape), pool[off:off + qn].view(qshape)
        run("ping-pong, q_alt = q + 64 GiB + %d B" % d)

#!/usr/bin/env python3
"""Round 4: does the time of the one-sweep kernel depend on the OFFSET of q inside its allocation (i.e. on the relative placement of the q streams
against x / p / x0, with the physical memory behind q unchanged)?  q is a view at `offset` bytes into one allocation of q + 2 GiB.
usage: python tools/q_offset_probe.py [--shape 256x8x1024x1024] [--offsets 0,4096,...]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x8x1024x1024")
ap.add_argument("--offsets", default="0,4096,65536,524288,1048576,2097152,4194304,8388608,34603008,67108864,134217728,268435456,536870912,1073741824,1610612736")
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
shape = tuple(int(v) for v in args.shape.split("x"))
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, fused=True, tune_placement=False)
qshape, qn = tuple(cp.q.shape), cp.q.numel()
del cp.q
torch.cuda.empty_cache()
offs = [int(v) for v in args.offsets.split(",")]
pool = torch.zeros(qn + max(offs) // 4 + 1024, dtype=torch.float32, device=dev)
print("# pool %.1f GiB at %#x; x %#x x_alt %#x p %#x x0 %#x" % (pool.numel() * 4 / 2 ** 30, pool.data_ptr(), cp.x.data_ptr(), cp.x_alt.data_ptr(), cp.p.data_ptr(), cp.x0.data_ptr()), flush=True)


def run(tag):
    for _ in range(2):
        cp.step()
    cp.timing = []
    torch.cuda.synchronize()
    for _ in range(args.steps):
        cp.step()
    torch.cuda.synchronize()
    k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]
    cp.timing = None
    print(json.dumps({"q offset": tag, "sweep_even": round(float(np.mean(k1[0::2])), 3), "sweep_odd": round(float(np.mean(k1[1::2])), 3), "sweep_mean": round(float(np.mean(k1)), 3)}), flush=True)


for rep in range(2):
    for o in offs:
        cp.q = pool[o // 4:o // 4 + qn].view(qshape)
        run(o)

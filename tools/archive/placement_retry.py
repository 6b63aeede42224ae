#!/usr/bin/env python3
"""Round 4, verdict item 1: is the slow mode of the north-star iteration a property of the ALLOCATION, and does allocating again
(while the first state is still held, so that different physical memory is handed out) change it?

Sequence inside ONE process: state A; state B while A is held; free A, state C while B is held; free B, state D; ... each timed.
usage: python tools/placement_retry.py [n_states=5] [--steps 8]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
t_import = time.perf_counter()
import numpy as np, torch, pytv
from bench import synth_slab, gpu_state
t_import = time.perf_counter() - t_import
n_states = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = 8
shape = (256, 8, 1024, 1024)
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
print("# import %.1f s; x0 at %s" % (t_import, hex(x0.data_ptr())), flush=True)


def measure(tag):
    cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, fused=True)
    for _ in range(3):
        cp.step()
    cp.timing = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        cp.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]; k2 = [e[1].elapsed_time(e[2]) for e in cp.timing]
    cp.timing = None
    row = {"state": tag, "ms_per_it": round(dt * 1e3, 3), "sweep_even": round(float(np.mean(k1[0::2])), 3), "sweep_odd": round(float(np.mean(k1[1::2])), 3),
           "fixup": round(float(np.mean(k2)), 3), "ptr": {n: hex(getattr(cp, n).data_ptr()) for n in ("x", "x_alt", "p", "q")},
           "clocks": {k: v for k, v in gpu_state(0).items() if k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w")}}
    print(json.dumps(row), flush=True)
    return cp


held = measure("A (first allocation of the process)")
for k in range(1, n_states):
    new = measure("%s (allocated while %s was held)" % (chr(65 + k), chr(64 + k)))
    del held
    torch.cuda.empty_cache()
    held = new
# the surviving state once more, now alone
held.timing = []
for _ in range(steps):
    held.step()
torch.cuda.synchronize()
k1 = [e[0].elapsed_time(e[1]) for e in held.timing]
print(json.dumps({"state": "last state again, alone", "sweep_even": round(float(np.mean(k1[0::2])), 3), "sweep_odd": round(float(np.mean(k1[1::2])), 3)}), flush=True)

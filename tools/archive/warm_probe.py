#!/usr/bin/env python3
"""Per-iteration time series of the one-sweep CP iteration from a cold process (is the run-to-run spread a warm-up effect?).
usage: python tools/warm_probe.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
x0 = synth_slab((256, 8, 1024, 1024), 0, 256, dev)
cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0)
cp.timing = []
torch.cuda.synchronize()
for _ in range(n):
    cp.step()
torch.cuda.synchronize()
k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]
print("sweep ms per iteration:", " ".join("%.1f" % v for v in k1))

#!/usr/bin/env python3
"""Run the one-sweep CP twice from the same input on the north-star shape: losses and final state must be bitwise
identical (fixed reduction trees, no atomics) -- a race in the in-block hand-off would break that under full load."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = (256, 8, 1024, 1024)
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
ref_loss, ref_x = None, None
for rep in range(4):
    cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0)
    loss = cp.run(8)
    x = cp.result()
    if ref_loss is None:
        ref_loss, ref_x = loss, x.clone()
    else:
        same = np.array_equal(loss, ref_loss) and torch.equal(x, ref_x)
        print("rep %d bitwise identical: %s" % (rep, same))
        if not same:
            print(loss, ref_loss); sys.exit(1)
    del cp
    torch.cuda.empty_cache()
print("deterministic")

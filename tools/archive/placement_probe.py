#!/usr/bin/env python3
"""Is the run-to-run spread of the one-sweep kernel (32.0 vs 34.5 ms on one box, same binary) a property of WHERE the arrays
land?  Re-allocates the solver state several times inside one process (optionally with a pad allocation in front, PAD_MB=...)
and prints the sweep time next to the low bits of the device pointers.
usage: python tools/placement_probe.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
shape = (256, 8, 1024, 1024)
dev = torch.device("cuda", 0)
pads = [int(v) for v in os.environ.get("PAD_MB", "0").split(",")]
for r in range(rounds):
    pad_mb = pads[r % len(pads)]
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
    x0 = synth_slab(shape, 0, shape[0], dev)
    cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0)
    for _ in range(3):
        cp.step()
    cp.timing = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        cp.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    k1 = np.mean([e[0].elapsed_time(e[1]) for e in cp.timing]); k2 = np.mean([e[1].elapsed_time(e[2]) for e in cp.timing])
    ptrs = {n: getattr(cp, n).data_ptr() for n in ("x", "q", "p", "x0") if hasattr(cp, n) and torch.is_tensor(getattr(cp, n))}
    print("round %d pad %4d MiB: %.2f ms/it sweep %.2f fixup %.2f  " % (r, pad_mb, dt * 1e3, k1, k2) +
          " ".join("%s=0x%x(mod 1GiB: %d MiB)" % (n, p, (p % (1 << 30)) >> 20) for n, p in ptrs.items()), flush=True)
    del cp, x0, pad
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Does any ADMM variant read memory it has not written?  The solver's scratch vectors come from torch.empty: the pool is
poisoned first (big tensors filled with a given value, then released), so a read-before-write shows up as a different (or
non-finite) trajectory.  usage: python tools/admm_determinism.py [NzxMxNyxNx]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (32, 16, 1024, 1024)
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
bad = 0
for scheme in ("upwind", "central", "hybrid"):
    for kw in (dict(x_solver="chebyshev"), dict(x_solver="chebyshev", fused=False), dict(), dict(fused=False)):
        res = []
        for poison in (0.0, 1e30, float("nan")):
            junk = [torch.full((x0.numel() * 5,), poison, device=dev) for _ in range(4)]
            del junk
            ad = pytv.solvers.ADMM(x0, 25.0, 0.05, n_cg=5, scheme=scheme, reg_time=1.0, **kw)
            loss = ad.run(3)
            res.append((loss.copy(), ad.result().clone()))
            del ad
        same = all(torch.equal(res[0][1], r[1]) for r in res[1:])
        bad += 0 if same else 1
        print(scheme, kw, "same x whatever the pool held:", same, [("%.9e" % r[0][-1]) for r in res], flush=True)
print("variants that read unwritten memory:", bad)
sys.exit(1 if bad else 0)

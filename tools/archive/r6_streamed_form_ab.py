"""Round 6: the streamed form of the persistent kernels (2 .. 4 site-vectors per thread, TV_SMALL_SITES) against the generic form it
replaces (TV_SMALL_SITES=1: resident or generic only) and the ordinary per-iteration path, on volumes whose site-vectors exceed 16 waves per CU."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from pytv import _native as nv
CASES = [((20, 4, 100, 100), torch.float64), ((40, 4, 100, 100), torch.float32), ((64, 4, 128, 128), torch.float32), ((16, 4, 256, 256), torch.float32),
         ((1, 1, 1024, 1024), torch.float32), ((1, 1, 512, 512), torch.float64)]
for shape, dt in CASES:
    x0 = (torch.rand(shape, dtype=torch.float64) * 100).to(dt).cuda()
    for scheme in ("hybrid", "upwind", "central"):
        for name, mk in (("CP", lambda p: pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0 if shape[1] > 1 else 0.0, persistent=p)),
                         ("SG", lambda p: pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, reg_time=1.0 if shape[1] > 1 else 0.0, persistent=p))):
            out, losses = [], []
            for p, sites in ((False, None), (True, 1), (True, None)):
                nv.set_option("TV_SMALL_SITES", sites)
                best = 1e9
                for rep in range(3):
                    s = mk(p); s.run(4); torch.cuda.synchronize(); t0 = time.perf_counter(); l = s.run(200); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                out.append(best / 200 * 1e6); losses.append(l)
                nv.set_option("TV_SMALL_SITES", None)
            rel = float(np.max(np.abs(losses[2] - losses[0]) / np.abs(losses[0])))
            print("%-16s %-8s %-8s %s  ordinary %7.2f | resident-or-generic %7.2f | with streamed %7.2f us/it | loss vs ordinary %.1e"
                  % ("x".join(map(str, shape)), str(dt).split(".")[1], scheme, name, out[0], out[1], out[2], rel), flush=True)

import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
np.random.seed(0)
x = np.random.rand(20, 4, 100, 100) * 100          # README.md:76-79: float64
for dt in (torch.float64, torch.float32):
    x0 = torch.as_tensor(x).to(dt).cuda()
    for scheme in ("hybrid", "upwind", "central"):
        for name, mk in (("CP", lambda p: pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0, persistent=p)),
                         ("SG", lambda p: pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, reg_time=1.0, persistent=p))):
            out = []
            for p in (False, True):
                best = 1e9
                for rep in range(3):
                    s = mk(p); s.run(4); torch.cuda.synchronize(); t0 = time.perf_counter(); l = s.run(300); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                out.append(best)
            print("%-8s %-8s %s  ordinary %6.2f us/it | persistent %6.2f us/it" % (str(dt).split(".")[1], scheme, name, out[0] / 300 * 1e6, out[1] / 300 * 1e6), flush=True)

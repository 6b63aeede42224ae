#!/usr/bin/env python3
"""Round 4, verdict item 7: which x-solve should ADMM default to?  50 outer iterations on the configs[4] per-GPU slab, four schemes,
rho in {0.02, 0.05, 0.2}: the primal objective 1/2 |x - x0|^2 + lambda |D x|_{2,1} against wall time for CG(5), Chebyshev(5),
Chebyshev(3) (and CG(3)), fp32 and -- on half the planes -- fp64.
usage: python tools/archive/admm_xsolve_study.py [NzxMxNyxNx=32x16x1024x1024] [n_outer=50] [--f64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab
args = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = tuple(int(v) for v in args[0].split("x")) if args else (32, 16, 1024, 1024)
n_outer = int(args[1]) if len(args) > 1 else 50
f64 = "--f64" in sys.argv
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
if f64:
    x0 = x0.double()
lam = 25.0
print("# ADMM x-solve study on %s %s, lambda = %g, %d outer iterations; objective = 1/2 |x - x0|^2 + lambda TV(x) of the iterate" % (
    shape, "fp64" if f64 else "fp32", lam, n_outer))
marks = [m for m in (5, 10, 20, 30, 50, 100) if m <= n_outer]
for scheme in ("upwind", "downwind", "central", "hybrid"):
    for rho in (0.02, 0.05, 0.2):
        res = {}
        for name, kw in (("cg5", dict(n_cg=5, x_solver="cg")), ("cheb5", dict(n_cg=5, x_solver="chebyshev")), ("cg3", dict(n_cg=3, x_solver="cg")), ("cheb3", dict(n_cg=3, x_solver="chebyshev")),
                         ("cheb8", dict(n_cg=8, x_solver="chebyshev"))):
            ad = pytv.solvers.ADMM(x0, lam, rho, scheme=scheme, reg_time=1.0, keep_z=False, **kw)
            ad.run(1)                                 # first outer iteration (no warm residual yet) + warm-up
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = ad.run(n_outer)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n_outer
            res[name] = (dt, loss)
            del ad
            torch.cuda.empty_cache()
        best = min(l[-1] for _, l in res.values())
        for name, (dt, loss) in res.items():
            # wall time until the objective is within 1e-3 / 1e-4 relative of the best final objective of the five variants
            def t_to(rel):
                idx = np.nonzero(loss <= best * (1.0 + rel))[0]
                return "%7.1f" % ((idx[0] + 1) * dt * 1e3) if len(idx) else "    n/a"
            print("%-8s rho %.2f %-6s %6.2f ms/outer | objective at %s: %s | ms to 1e-3: %s  to 1e-4: %s  | final / best - 1 = %.2e" % (
                scheme, rho, name, dt * 1e3, marks, " ".join("%.6e" % loss[m - 1] for m in marks), t_to(1e-3), t_to(1e-4), loss[-1] / best - 1.0), flush=True)

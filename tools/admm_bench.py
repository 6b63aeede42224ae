#!/usr/bin/env python3
"""ADMM outer iterations/s (BASELINE config 4 flavour: M = 16, all four discretisations) on one GPU.
usage: python tools/admm_bench.py [NzxMxNyxNx] [n_cg]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (32, 16, 1024, 1024)
n_cg = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
V = x0.numel()
print("ADMM on %s (V = %.0f Mvox), %d CG steps per outer iteration, rho = 0.05, lambda = 25" % (shape, V / 1e6, n_cg))
for scheme in ("upwind", "downwind", "central", "hybrid"):
    # one-sweep dual side (round 3, sparse / full storage of t'), the kernel trio it replaces, the textbook recurrence
    for name, kw in (("default", dict()),          # what ADMM(x0, reg, rho) is since round 4: one-sweep, Chebyshev x-solve, keep_z=True, placement tuner
                     ("one-sweep+chebyshev", dict(fused=True, x_solver="chebyshev", keep_z=False)), ("one-sweep", dict(fused=True, keep_z=False, x_solver="cg")),
                     ("one-sweep keep_z", dict(fused=True, keep_z=True, x_solver="cg")),
                     ("single-reduction", dict(fused=False, x_solver="cg")), ("textbook CG", dict(single_reduction=False))):
        if os.environ.get("ONLY") and os.environ["ONLY"] != name:          # ONLY=<exact name>: one variant (profiling)
            continue
        ad = pytv.solvers.ADMM(x0, 25.0, 0.05, n_cg=n_cg, scheme=scheme, reg_time=1.0, **kw)
        ad.run(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = ad.run(4)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        nd = ad.geo.nd
        # words / voxel per outer iteration.  single reduction: rhs (Nd + 2) + residual (3) + n_cg normal ops on r (2) + n_cg
        # updates (9; the first 7; the last +1 for x0) + t/u update (1 + 3 Nd) = 4 Nd + 11 n_cg + 5; textbook: rhs (2 Nd + 2) +
        # residual with the copy (4) + n_cg (normal op 2, cg1 6, cg2 3) + z/u (1 + 3 Nd) + fidelity (sub 3 + dot 2);
        # one-sweep: sweep (x, x0 read, r written, u read + written: 2 Nd + 3; + Nd with keep_z) + n_cg (2 + 9) - 1
        # + chebyshev: the x-solve is e_2 from r alone (2), a step with y = a0 r formed on the fly (3), n_cg - 4 steps of 4 (e_k with its
        # stencil, r, e_{k-1} read; e_{k+1} written) and a last step of 5 (+ x read; the new x written instead of e; |x - x0|^2 comes
        # from the sweep that follows): 4 n_cg - 6 words
        words = {"default": 3 * nd + 3 + 4 * n_cg - 6, "one-sweep+chebyshev": 2 * nd + 3 + 4 * n_cg - 6, "one-sweep": 2 * nd + 11 * n_cg + 2, "one-sweep keep_z": 3 * nd + 11 * n_cg + 2, "single-reduction": 4 * nd + 11 * n_cg + 5,
                 "textbook CG": 5 * nd + 11 * n_cg + 12}[name]
        print("%-9s Nd=%d %-19s %.2f ms/outer  %.1f it/s  loss %.6e -> %.6e  | algorithmic %.0f words/voxel -> %.0f GB/s" % (
            scheme, nd, name, dt * 1e3, 1 / dt, loss[0], loss[-1], words, words * 4.0 * V / dt / 1e9))
        if ad.placement:
            print("          placement tuner: %s" % (ad.placement,))
        del ad
        torch.cuda.empty_cache()

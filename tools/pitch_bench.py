#!/usr/bin/env python3
"""Round 4, verdict item 4: what padded solver state (tv_geom::row_pitch / frame_pitch) buys on frames whose rows are not whole
cache lines -- the reference's own shapes (pytv/tests.py:48 N = 100, README.md:76-79 rand(20, 4, 100, 100)) and CT-like 1000 / 1001
column frames.  Dense vs pitch="auto", same process, interleaved.
usage: python tools/pitch_bench.py [shapes ...]   default: 64x8x1000x1000 64x8x1001x1001 64x8x1024x1024 256x4x100x100"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from pytv import _native as nv
from bench import synth_slab

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 8, 1000, 1000), (64, 8, 1001, 1001), (64, 8, 1024, 1024), (256, 4, 100, 100)]
dev = torch.device("cuda", 0)
lib = nv.lib()


def timed(fn, n=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for shape in shapes:
    V = float(np.prod(shape))
    x0 = synth_slab(shape, 0, shape[0], dev)
    print("== %s  (%.0f Mvox)" % ("x".join(map(str, shape)), V / 1e6), flush=True)
    for scheme in ("hybrid", "upwind"):
        rows = {}
        for rep in range(2):
            for name, pitch in (("dense", None), ("pitched", "auto")):
                # ---- Chambolle-Pock iteration ------------------------------------------------------------------------------
                cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0, pitch=pitch)
                t_cp = timed(cp.step, n=6)
                nd, fused = cp.geo.nd, cp.fused
                words_cp = (5 + 2 * nd) if fused else (6 + 3 * nd)
                geo = cp.geo
                # ---- tv_D through the C-ABI on the solver's own arrays --------------------------------------------------------
                st = nv.current_stream(dev)
                t_D = timed(lambda: nv.check(lib.tv_D(geo.ref, nv.ptr(cp.x), None, None, nv.ptr(cp.q), st)))
                del cp
                torch.cuda.empty_cache()
                # ---- sub-gradient descent step -------------------------------------------------------------------------------
                sg = pytv.solvers.SubgradientDescent(x0, 25.0, 0.02, scheme=scheme, reg_time=1.0, pitch=pitch)
                t_sg = timed(sg.step, n=6)
                one_pass = sg.one_pass
                del sg
                torch.cuda.empty_cache()
                # ---- ADMM outer iteration, Chebyshev x-solve, 5 steps -----------------------------------------------------------
                ad = pytv.solvers.ADMM(x0, 25.0, 0.05, n_cg=5, scheme=scheme, reg_time=1.0, x_solver="chebyshev", keep_z=False, pitch=pitch)
                sc = torch.zeros(2, dtype=torch.float64, device=dev)
                t_ad = timed(lambda: ad.step(sc), n=4)
                ad_fused = ad.fused
                del ad
                torch.cuda.empty_cache()
                rows.setdefault(name, []).append((t_cp, t_D, t_sg, t_ad, fused, one_pass, ad_fused, words_cp, nd, geo.row_pitch, geo.frame_pitch))
        for name, r in rows.items():
            t_cp, t_D, t_sg, t_ad = (min(v[i] for v in r) for i in range(4))
            _, _, _, _, fused, one_pass, ad_fused, words_cp, nd, rp, fp = r[0]
            print("  %-7s %-8s rp %5d fp %8d | CP %s %7.3f ms (%.2f of %d words) | tv_D %6.3f ms (%.2f of %d words) | sub-gradient step %s %6.3f ms (%.2f of 3 words) | ADMM+cheb %s %7.3f ms"
                  % (scheme, name, rp, fp, "one-sweep" if fused else "pair     ", t_cp, words_cp * 4 * V / t_cp / 1e6 / 8000, words_cp, t_D,
                     (1 + nd) * 4 * V / t_D / 1e6 / 8000, 1 + nd, "one-pass" if one_pass else "two-pass", t_sg, 3 * 4 * V / t_sg / 1e6 / 8000,
                     "one-sweep" if ad_fused else "trio", t_ad), flush=True)
    del x0
    torch.cuda.empty_cache()

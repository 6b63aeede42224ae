"""Residency extremes of the register-resident persistent kernels: many tiny frames = many small blocks per CU.  A launch whose blocks are not all
resident abandons itself (NaN history -> RuntimeError) instead of hanging; this prints what each extreme does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
for shape in ((64, 8, 16, 16), (256, 8, 16, 16), (512, 8, 16, 16), (512, 8, 8, 8), (1024, 8, 8, 8), (2048, 2, 8, 8), (100, 16, 20, 20), (6, 3, 700, 400), (1, 1, 2048, 512)):
    x0 = torch.rand(shape, device="cuda") * 100
    for scheme in ("hybrid", "upwind"):
        t0 = time.perf_counter()
        try:
            cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0, persistent=True)
            ref = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0, fused=False)
            l = cp.run(20); lr = ref.run(20, graph=False)
            sg = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, reg_time=1.0, persistent=True)
            ls = sg.run(9)
            ok = np.allclose(l, lr, rtol=2e-5) and np.all(np.isfinite(ls))
            print("%-20s %-7s %s  CP loss %.6e (kernel pair %.6e)  %.2f s" % ("x".join(map(str, shape)), scheme, "ok" if ok else "MISMATCH", l[-1], lr[-1], time.perf_counter() - t0), flush=True)
        except Exception as e:
            print("%-20s %-7s ERROR %s (%.2f s)" % ("x".join(map(str, shape)), scheme, str(e)[:120], time.perf_counter() - t0), flush=True)

#!/usr/bin/env python3
"""tv_normal_op2 (streaming kernel) against the oracle and against the one-site kernels, and the two ADMM recurrences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from pytv import _native as nv
from oracle import tv_oracle as orc
nv.set_option("TV_MARCH_MIN_PLANE_KB", 0)
lib = nv.lib()
rng = np.random.default_rng(0)
bad = 0
for shape in [(7, 3, 9, 256), (5, 1, 6, 128), (6, 2, 5, 132), (9, 8, 6, 192), (3, 16, 5, 128), (4, 12, 7, 64), (1, 1, 33, 68), (1, 4, 8, 64),
              (8, 5, 3, 64), (1, 20, 2, 72), (11, 1, 1, 260)]:
    for scheme in ("upwind", "hybrid", "downwind", "central"):
        if scheme == "central" and (shape[0] == 2 or shape[1] == 2):
            continue
        for zc in (0, 3):
            nv.set_option("TV_ZCHUNK", zc)
            kw = dict(reg_z_over_reg=1.3, reg_time=0.5)
            x = torch.as_tensor((rng.standard_normal(shape) * 10).astype(np.float32)).cuda()
            b = torch.as_tensor((rng.standard_normal(shape) * 10).astype(np.float32)).cuda()
            g = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
            st, ws = nv.current_stream(x.device), g.workspace()
            x64 = x.double().cpu().numpy()
            want = x64 + 0.3 * orc.D_T(orc.D(x64, scheme, **kw), scheme, **kw)
            for mode in ("plain", "residual"):
                out, out2 = torch.empty_like(x), torch.empty_like(x)
                dots = torch.zeros(2, dtype=torch.float64, device="cuda")
                for kern in (2, 0):
                    nv.set_option("TV_NORMAL_KERNEL", kern)
                    nv.check(lib.tv_normal_op2(g.ref, nv.ptr(x), None, None, 0.3, nv.ptr(b) if mode == "residual" else None, nv.ptr(out),
                                               nv.ptr(out2) if mode == "residual" else None, dots.data_ptr(), nv.ptr(ws), st))
                    w = want if mode == "plain" else b.double().cpu().numpy() - want
                    err = np.abs(out.cpu().numpy() - w).max()
                    d0 = float(np.sum(x64 * want)) if mode == "plain" else float(np.sum(w * w))
                    d1 = float(np.sum(x64 * x64))
                    e0, e1 = abs(dots[0].item() - d0) / abs(d0), abs(dots[1].item() - d1) / d1
                    ok = err < 2e-3 and e0 < 1e-5 and e1 < 1e-6 and (mode == "plain" or torch.equal(out, out2))
                    bad += (not ok)
                    if not ok:
                        print("MISMATCH", shape, scheme, "zchunk", zc, mode, "kernel", kern, "max err %.3e dots %.2e %.2e" % (err, e0, e1))
nv.set_option("TV_NORMAL_KERNEL", None)
nv.set_option("TV_ZCHUNK", None)
for shape in [(7, 3, 9, 256), (5, 3, 8, 12)]:
    for scheme in ("upwind", "hybrid", "central"):
        x0 = (50.0 * rng.random(shape)).astype(np.float32)
        for single in (True, False):
            ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), 5.0, 0.1, n_cg=4, scheme=scheme, reg_time=0.5, single_reduction=single, x_solver="cg")
            loss = ad.run(4)
            wx, wloss = orc.admm(x0.astype(np.float64), 4, 5.0, 0.1, 4, scheme=scheme, reg_time=0.5, single_reduction=single, x_solver="cg")
            rel = np.abs(loss - wloss).max() / wloss.max()
            ok = rel < 2e-5
            bad += (not ok)
            print("ADMM", shape, scheme, "single" if single else "textbook", "rel loss err %.2e" % rel, "ok" if ok else "MISMATCH")
print("failures:", bad)

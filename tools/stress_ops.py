#!/usr/bin/env python3
"""Randomised cross-check of every entry point against the CPU oracle over M = 1..20 and plane sizes on both sides of the
marching threshold (the dispatch has many (scheme, M, size) branches: this walks them).  usage: python tools/stress_ops.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"
import numpy as np, torch, pytv
from pytv import _native as nv
from oracle import tv_oracle as orc
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "99")))
bad = 0
TOL = [None]
def check(name, got, want, info, **kw):
    global bad
    if not np.allclose(got, want, **(kw or TOL[0])):
        bad += 1
        print("MISMATCH", name, info, float(np.abs(np.asarray(got) - np.asarray(want)).max()))
for case in range(n_cases):
    scheme = ["upwind", "downwind", "hybrid", "central"][case % 4]
    m = int(rng.integers(1, 21))
    nz = int(rng.choice([1, 2, 3, 5, 9, 18, 35]))
    ny = int(rng.integers(2, 40))
    nx = 4 * int(rng.choice([3, 16, 17, 32, 33, 64, 65, 130]))
    while nz * m * ny * nx > 400000:          # keep the NumPy oracle fast
        nz = max(1, nz // 2); ny = max(2, ny // 2)
    os.environ["TV_ZCHUNK"] = str(int(rng.choice([0, 2, 3, 16])))
    nv.set_option("TV_ZCHUNK", int(os.environ["TV_ZCHUNK"]))
    lz = float(rng.choice([0.0, 1.0, 2.5])); mu = float(rng.choice([0.0, 0.5, 1.0]))
    if scheme == "central" and (nz == 2 or m == 2):
        nz, m = 3, max(m, 3)
    kw = dict(reg_z_over_reg=lz, reg_time=mu)
    kind = int(rng.integers(0, 5))             # 0, 1: plain; 2: boolean mask; 3: per-pixel weight map; 4: per-voxel weight volume
    if kind == 2:
        kw.update(mask_static=rng.random((ny, nx)) < 0.4, factor_reg_static=2.3)
    elif kind == 3:
        kw.update(mask_static=rng.random((ny, nx)) * 2.0)
    elif kind == 4:
        kw.update(mask_static=rng.random((nz, m, ny, nx)) * 2.0)
    info = (scheme, (nz, m, ny, nx), lz, mu, ("plain", "plain", "mask", "weights", "weight volume")[kind], os.environ["TV_ZCHUNK"])
    dt = np.float64 if rng.random() < 0.25 else np.float32          # fp64: the one-site-per-thread kernels, any M
    tol = dict(rtol=1e-5, atol=2e-5) if dt == np.float32 else dict(rtol=1e-10, atol=1e-9)
    TOL[0] = tol
    lt = 1.0 if dt == np.float32 else 1e-5                          # scale of the loss tolerances
    info = info + (np.dtype(dt).name,)
    x = (rng.standard_normal((nz, m, ny, nx)) * 10).astype(dt)
    x64 = x.astype(np.float64)
    ops, tvg = pytv.tv_operators_GPU, pytv.tv_GPU
    d = getattr(ops, "D_" + scheme)(x, **kw)
    check("D", d, orc.D(x64, scheme, **kw), info)
    y = rng.standard_normal(d.shape).astype(dt)
    check("DT", getattr(ops, "D_T_" + scheme)(y, **kw), orc.D_T(y.astype(np.float64), scheme, **kw), info)
    wtv, wG = orc.tv(x64, scheme, **kw)
    for norms in (True, False):
        out = getattr(tvg, "tv_" + scheme)(x.copy(), return_grad_norms=norms, **kw)
        check("tv(norms=%s)" % norms, float(out[0]), wtv, info, rtol=1e-6 * lt, atol=0)
        check("G(norms=%s)" % norms, out[1], wG, info)
    x0 = torch.as_tensor(x * 5).cuda()
    rx, rl = orc.chambolle_pock(x64 * 5, 4, 7.0, scheme=scheme, **kw)
    for fused in (False, None):
        cp = pytv.solvers.ChambollePock(x0, 7.0, scheme=scheme, fused=fused, **kw)
        check("cp loss fused=%s" % cp.fused, cp.run(4), rl, info, rtol=1e-5 * lt, atol=0)
        check("cp x fused=%s" % cp.fused, cp.result().cpu().numpy(), rx, info, rtol=1e-4 * lt, atol=1e-3 * lt)
    for single in (True, False):
        ad = pytv.solvers.ADMM(x0, 7.0, 0.1, n_cg=3, scheme=scheme, single_reduction=single, x_solver="cg", **kw)
        _, al = orc.admm(x64 * 5, 2, 7.0, 0.1, 3, scheme=scheme, single_reduction=single, **kw)
        check("admm single=%s" % single, ad.run(2), al, info, rtol=1e-4 * lt, atol=0)
    sg = pytv.solvers.SubgradientDescent(x0, 2.0, 0.02, scheme=scheme, **kw)
    _, sl = orc.subgradient_descent(x64 * 5, 3, 2.0, 0.02, scheme=scheme, **kw)
    check("sg", sg.run(3), sl, info, rtol=1e-4 * lt, atol=0)
print("cases %d, mismatches %d" % (n_cases, bad))
sys.exit(1 if bad else 0)

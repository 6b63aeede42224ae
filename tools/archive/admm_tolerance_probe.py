import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import pytv
from oracle import tv_oracle_c as occ
from oracle import tv_oracle as orc
shape, n_outer, n_cg = (64, 16, 64, 1024), 2, 3
kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
rng = np.random.default_rng(33)
x0 = (60.0 * rng.random(shape)).astype(np.float32)
for scheme in ("upwind", "downwind", "central", "hybrid"):
    wx, wloss, wz, wu = occ.admm(x0.astype(np.float64), n_outer, 7.0, 0.1, n_cg, scheme=scheme, return_state=True, single_reduction=True, **kw)
    for fused in (True, False):
        ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), 7.0, 0.1, n_cg=n_cg, scheme=scheme, keep_z=True, fused=fused, x_solver="cg", **kw)
        loss = ad.run(n_outer)
        x = ad.result().cpu().numpy().astype(np.float64); z = ad.z.cpu().numpy().astype(np.float64)
        print("config4-like %-8s fused=%d loss rel %.2e  x abs %.2e (max |x| %.1f)  z abs %.2e (max |z| %.1f)" % (
            scheme, fused, np.max(np.abs(loss - wloss) / np.abs(wloss)), np.max(np.abs(x - wx)), np.abs(wx).max(), np.max(np.abs(z - wz)), np.abs(wz).max()))
# the small oracle test of test_gpu_parity.py / test_gpu_admm_fused.py
for shape, lz, mu in (((1, 1, 24, 64), 1.0, 0.0), ((5, 3, 16, 64), 1.5, 0.5), ((3, 10, 9, 128), 1.0, 0.7), ((5, 3, 8, 12), 1.5, 0.5)):
    rng = np.random.default_rng(6)
    x0 = (rng.random(shape) * 100).astype(np.float32)
    for scheme in ("upwind", "downwind", "central", "hybrid"):
        wx, wloss, wz, wu = orc.admm(x0.astype(np.float64), 6, 25.0, 0.05, 5, scheme=scheme, reg_z_over_reg=lz, reg_time=mu, single_reduction=True, return_state=True)
        ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), 25.0, 0.05, n_cg=5, scheme=scheme, reg_z_over_reg=lz, reg_time=mu, keep_z=True, x_solver="cg")
        loss = ad.run(6)
        x = ad.result().cpu().numpy().astype(np.float64); z = ad.z.cpu().numpy().astype(np.float64); u = ad.u.cpu().numpy().astype(np.float64)
        print("small %s %-8s fused=%d loss rel %.2e  x abs %.2e  z abs %.2e  u abs %.2e (max |x| %.0f |z| %.0f)" % (
            shape, scheme, ad.fused, np.max(np.abs(loss - wloss) / np.abs(wloss)), np.max(np.abs(x - wx)), np.max(np.abs(z - wz)), np.max(np.abs(u - wu)), np.abs(wx).max(), np.abs(wz).max()))

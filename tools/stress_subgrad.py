#!/usr/bin/env python3
"""Soak test of the one-pass TV + sub-gradient kernel: random shapes / weights / masks / z-chunks, compared with the
two-pass kernels (IEEE arithmetic) and run twice for bitwise determinism.  usage: python tools/stress_subgrad.py [n_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"
import numpy as np, torch, pytv
from pytv import _native as nv
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "2024")))
bad = done = 0
for case in range(n_cases):
    scheme = ["upwind", "downwind", "hybrid", "central"][case % 4]
    m = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 16, 19]))
    nz = int(rng.integers(1, 12))
    ny = int(rng.integers(1, 70))
    nx = 4 * int(rng.integers(1, 80)) if case % 3 else int(rng.integers(3, 320))     # every third case: any Nx (late round 3)
    lz = float(rng.choice([0.0, 0.3, 1.0, 2.5])); mu = float(rng.choice([0.0, 2.0 ** -5, 1.0, 1.7]))
    use_mask = bool(rng.random() < 0.3)
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=(rng.random((ny, nx)) < 0.4) if use_mask else False,
              factor_reg_static=2.3 if use_mask else 0)
    os.environ["TV_ZCHUNK"] = str(int(rng.choice([0, 1, 2, 3, 5, 16])))
    nv.set_option("TV_ZCHUNK", int(os.environ["TV_ZCHUNK"]))
    x = torch.as_tensor((rng.standard_normal((nz, m, ny, nx)) * 10).astype(np.float32)).cuda()
    geo = nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, **kw)
    if not nv.lib().tv_subgrad_fused_supported(geo.ref):
        continue
    tv1, G1, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=False, one_pass=True, **kw)
    tv1b, G1b, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=False, one_pass=True, **kw)
    tv2, G2, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, one_pass=False, **kw)
    ok = torch.equal(G1, G1b) and float(tv1) == float(tv1b) and torch.allclose(G1, G2, rtol=1e-5, atol=2e-5) \
        and abs(float(tv1) - float(tv2)) <= 1e-6 * max(1.0, abs(float(tv2)))
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH", scheme, tuple(x.shape), kw["reg_z_over_reg"], kw["reg_time"], use_mask, os.environ["TV_ZCHUNK"],
              float((G1 - G2).abs().max()), float(tv1), float(tv2))
print("cases run %d, mismatches %d" % (done, bad))
sys.exit(1 if bad else 0)

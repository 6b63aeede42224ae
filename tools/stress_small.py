#!/usr/bin/env python3
"""Soak test of the persistent small-volume loops (tv_small_cp / tv_small_subgrad_descent): random shapes / schemes / dtypes / weights /
masks / pitches / iteration counts, the register-resident, streamed and generic forms, against the ordinary per-iteration kernels (kernel pair;
two-pass sub-gradient + step) on the same state, and run twice for determinism.  usage: python tools/stress_small.py [n_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from pytv import _native as nv
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "606")))
bad = done = 0
for case in range(n_cases):
    scheme = ["upwind", "downwind", "hybrid", "central"][case % 4]
    m = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 9]))
    nz = int(rng.integers(1, 9))
    ny = int(rng.integers(2, 60))
    nx = 4 * int(rng.integers(1, 40)) if case % 3 else int(rng.integers(3, 150))         # every third case: any Nx
    lz = float(rng.choice([0.0, 0.3, 1.0, 2.5])); mu = float(rng.choice([0.0, 2.0 ** -5, 1.0, 1.7]))
    use_mask = bool(rng.random() < 0.3)
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=(rng.random((ny, nx)) < 0.4) if use_mask else False, factor_reg_static=2.3 if use_mask else 0)
    dtype = np.float64 if case % 5 == 0 else np.float32
    pitch = [None, "auto"][int(rng.integers(0, 2))]
    form = str(rng.choice(["registers", "registers", "generic", "generic", "streamed2", "streamed3", "streamed4"]))
    generic = form == "generic"
    n_it = int(rng.choice([1, 2, 7, 16, 33]))
    x0 = torch.as_tensor((rng.standard_normal((nz, m, ny, nx)) * 30 + 50).astype(dtype)).cuda()
    if os.environ.get("STRESS_VERBOSE"):
        print("case", case, scheme, tuple(x0.shape), dtype.__name__, lz, mu, use_mask, pitch, form, n_it, flush=True)
    nv.set_option("TV_SMALL_GENERIC", 1 if generic else None)
    nv.set_option("TV_SMALL_SITES", int(form[-1]) if form.startswith("streamed") else None)
    try:
        tol = 1e-9 if dtype == np.float64 else 2e-5
        a = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, persistent=True, pitch=pitch, **kw)
        a2 = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, persistent=True, pitch=pitch, **kw)
        b = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, fused=False, pitch=pitch, **kw)
        la, la2, lb = a.run(n_it), a2.run(n_it), b.run(n_it, graph=False)
        xa, xb = a.result(), b.result()
        ok = np.array_equal(la, la2) and torch.equal(xa, a2.result()) and np.allclose(la, lb, rtol=tol) and torch.allclose(xa, xb, rtol=tol, atol=tol * 100)
        s = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, persistent=True, pitch=pitch, **kw)
        s2 = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, persistent=True, pitch=pitch, **kw)
        t = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3, scheme=scheme, one_pass=False, pitch=pitch, **kw)
        ls, ls2, lt = s.run(n_it), s2.run(n_it), t.run(n_it, graph=False)
        # a descent loop amplifies rounding differences at kinks: compare the head tightly, the rest loosely
        h = min(n_it, 5)
        ok2 = np.array_equal(ls, ls2) and torch.equal(s.result(), s2.result()) and np.allclose(ls[:h], lt[:h], rtol=tol) and np.allclose(ls, lt, rtol=max(tol, 1e-4 if dtype == np.float32 else 1e-8))
    finally:
        nv.set_option("TV_SMALL_GENERIC", None)
        nv.set_option("TV_SMALL_SITES", None)
    done += 1
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", scheme, tuple(x0.shape), dtype.__name__, lz, mu, use_mask, pitch, form, n_it, "CP" if not ok else "", "SG" if not ok2 else "",
              float(np.max(np.abs(la - lb) / np.abs(lb))), float(np.max(np.abs(ls - lt) / np.abs(lt))))
print("cases run %d, mismatches %d" % (done, bad))
sys.exit(1 if bad else 0)

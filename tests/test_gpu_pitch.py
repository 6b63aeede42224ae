"""Pitched arrays (tv_geom::row_pitch / frame_pitch, interface version 4) on the GPU: every pitch-aware entry point must give,
on the valid region, exactly what it gives on dense arrays -- and what the oracle gives -- and must leave the pads zero.
The reference has no pitch (pytv/tv_operators_CPU.py:82-83: dense (Nz, M, N, N)); its own canonical shapes (pytv/tests.py:48
N = 100; README.md:76-79 rand(20, 4, 100, 100)) are what the padded solver state is for."""
import os

import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"


@pytest.fixture(scope="module")
def pytv():
    import pytv
    return pytv


def _x(shape, dtype, seed=0):
    rng = np.random.default_rng(seed)
    return (orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(dtype)


def _pads_zero(t):
    """every storage element of the strided view t that is NOT one of its elements is zero"""
    import torch
    n = 1 + sum((int(s) - 1) * int(st) for s, st in zip(t.shape, t.stride()))
    flat = t.as_strided((n,), (1,))
    total = flat.double().abs().sum().item()
    inside = t.double().abs().sum().item()
    return abs(total - inside) <= 1e-9 * max(1.0, total)


PITCHES = [
    ((6, 3, 32, 64), (80, 32 * 80 + 12)),        # row pad + frame pad
    ((5, 2, 16, 128), (128, 16 * 128 + 1088)),   # frame pad only (the north-star layout in small)
    ((9, 8, 24, 256), (260, 24 * 260)),          # row pad only
    ((4, 11, 16, 64), (64, 16 * 64 + 4)),        # time windows (M > 8)
]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,pitch", PITCHES)
def test_one_sweep_cp_pitched_equals_dense_and_oracle(pytv, scheme, dtype, shape, pitch):
    import torch
    x0 = _x(shape, dtype)
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    n = 6
    dense = pytv.solvers.ChambollePock(torch.as_tensor(x0).cuda(), 20.0, scheme=scheme, fused=True, **kw)
    ld = dense.run(n)
    pit = pytv.solvers.ChambollePock(torch.as_tensor(x0).cuda(), 20.0, scheme=scheme, fused=True, pitch=pitch, **kw)
    assert pit.geo.pitched and pit.x.stride()[-2] == pitch[0] and pit.x.stride()[-3] == pitch[1]
    lp = pit.run(n)
    # same blocks, same arithmetic: the iterates are bit-identical; the loss agrees to fp64 rounding (the last fidelity of a block of
    # iterations is a flat reduction over the storage, whose partial sums group the padded array differently)
    np.testing.assert_allclose(lp, ld, rtol=1e-14)
    assert torch.equal(pit.result(), dense.result())
    assert torch.equal(pit.q, dense.q)
    for t in (pit.x, pit.x_alt, pit.p, pit.q):
        assert _pads_zero(t)
    _, wloss = orc.chambolle_pock(x0.astype(np.float64), n, 20.0, scheme=scheme, **kw)
    np.testing.assert_allclose(lp, wloss, rtol=1e-5 if dtype == np.float32 else 1e-11)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape,pitch", PITCHES[:3] + [((3, 2, 9, 14), (16, 9 * 16 + 8)), ((4, 3, 7, 13), (16, 7 * 16))])
def test_kernel_pair_cp_pitched(pytv, scheme, dtype, shape, pitch):
    """tv_cp_dual + tv_cp_primal (one-site kernels when the state is pitched) on padded state == oracle; ragged Nx (13, 14 columns)
    runs through the scalar lanes of the same kernels"""
    import torch
    x0 = _x(shape, dtype, 1)
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    n = 5
    pit = pytv.solvers.ChambollePock(torch.as_tensor(x0).cuda(), 20.0, scheme=scheme, fused=False, pitch=pitch, **kw)
    lp = pit.run(n)
    wx, wloss = orc.chambolle_pock(x0.astype(np.float64), n, 20.0, scheme=scheme, **kw)
    tol = dict(rtol=1e-5, atol=1e-3) if dtype == np.float32 else dict(rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lp, wloss, rtol=tol["rtol"])
    np.testing.assert_allclose(pit.result().cpu().numpy(), wx, **tol)
    for t in (pit.x, pit.p, pit.q):
        assert _pads_zero(t)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("variant", ["fused-cg", "fused-cheb", "trio", "textbook"])
@pytest.mark.parametrize("shape,pitch", [((6, 3, 16, 64), (72, 16 * 72 + 20)), ((5, 3, 12, 71), (72, 12 * 72 + 8))])
def test_admm_pitched(pytv, scheme, dtype, variant, shape, pitch):
    """ADMM on padded state; 71 columns on a 72-element pitch: ragged rows on 16-byte lanes in the one-sweep kernels AND in the streaming
    normal operator / Chebyshev step (round 4: the column differences that touch pad columns are masked)"""
    import torch
    x0 = _x(shape, dtype, 2)
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    okw = dict(single_reduction=(variant != "textbook"), x_solver="chebyshev" if variant == "fused-cheb" else "cg")
    skw = dict(okw, fused=variant.startswith("fused"))
    ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), 4.0, 0.1, n_cg=4, scheme=scheme, pitch=pitch, **skw, **kw)
    assert ad.fused == variant.startswith("fused") and ad.geo.pitched
    loss = ad.run(4)
    wx, wloss, wz, wu = orc.admm(x0.astype(np.float64), 4, 4.0, 0.1, 4, scheme=scheme, return_state=True, **okw, **kw)
    tol = dict(rtol=2e-5, atol=2e-3) if dtype == np.float32 else dict(rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(loss, wloss, rtol=1e-5 if dtype == np.float32 else 1e-10)
    np.testing.assert_allclose(ad.result().cpu().numpy(), wx, **tol)
    np.testing.assert_allclose(ad.z.cpu().numpy(), wz, **tol)
    for t in (ad.x, ad.u, ad._zt, ad.r):
        assert _pads_zero(t)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_subgradient_descent_pitched(pytv, scheme, dtype):
    import torch
    shape, pitch = (5, 3, 12, 30), (32, 12 * 32 + 8)
    x0 = _x(shape, dtype, 3)
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    sg = pytv.solvers.SubgradientDescent(torch.as_tensor(x0).cuda(), 2.0, 0.05, scheme=scheme, pitch=pitch, **kw)
    loss = sg.run(5)
    wx, wloss = orc.subgradient_descent(x0.astype(np.float64), 5, 2.0, 0.05, scheme=scheme, **kw)
    np.testing.assert_allclose(loss, wloss, rtol=2e-5 if dtype == np.float32 else 1e-10)
    np.testing.assert_allclose(sg.result().cpu().numpy(), wx, rtol=1e-4 if dtype == np.float32 else 1e-9, atol=1e-3 if dtype == np.float32 else 1e-9)
    assert _pads_zero(sg.x)


def test_bad_pitches_are_argument_errors(pytv):
    import torch
    from pytv import _native as nv
    for rp, fp in ((60, 0), (66, 0), (0, 16 * 64 - 4), (0, 16 * 64 + 2)):
        with pytest.raises(ValueError):
            nv.Geometry((4, 2, 16, 64), "hybrid", torch.float32, "cuda", reg_time=1.0, row_pitch=rp, frame_pitch=fp)

"""The C / OpenMP oracle (oracle/tv_oracle_c.c) against the NumPy oracle, which is pinned to the reference's golden
vectors.  CPU only."""
import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc
from oracle import tv_oracle_c as occ

GEOMS = [((1, 1, 9, 11), 1.0, 0.0, False), ((5, 1, 8, 8), 1.0, 0.0, False), ((4, 3, 7, 6), 2.5, 0.7, True), ((2, 2, 6, 9), 1.0, 1.0, False),
         ((6, 8, 5, 5), 0.0, 2 ** -5, False), ((3, 4, 6, 6), 1.3, 0.0, False)]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_c_oracle_operators_equal_numpy_oracle(scheme, dtype):
    rng = np.random.default_rng(3)
    tol = dict(rtol=1e-12, atol=1e-12) if dtype == np.float64 else dict(rtol=2e-6, atol=2e-6)
    for shape, lz, mu, use_mask in GEOMS:
        mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=3.0 if use_mask else 0)
        x = rng.standard_normal(shape).astype(dtype)
        want = orc.D(x.astype(np.float64), scheme, **kw)
        got = occ.D(x, scheme, **kw)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, err_msg="D %s %s" % (scheme, shape), **tol)
        y = rng.standard_normal(want.shape).astype(dtype)
        np.testing.assert_allclose(occ.D_T(y, scheme, **kw), orc.D_T(y.astype(np.float64), scheme, **kw),
                                   err_msg="DT %s %s" % (scheme, shape), **tol)
        np.testing.assert_allclose(occ.compute_L21_norm(got, shape), orc.compute_L21_norm(want), rtol=tol["rtol"] * 10)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_c_oracle_chambolle_pock_equals_numpy_oracle(scheme):
    rng = np.random.default_rng(4)
    for shape, lz, mu, use_mask in GEOMS[:4]:
        mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=3.0 if use_mask else 0)
        x0 = 50 * rng.random(shape)
        wx, wloss = orc.chambolle_pock(x0, 15, 5.0, scheme=scheme, **kw)
        gx, gloss = occ.chambolle_pock(x0, 15, 5.0, scheme=scheme, **kw)
        np.testing.assert_allclose(gloss, wloss, rtol=1e-11)
        np.testing.assert_allclose(gx, wx, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("single", [False, True])
def test_c_oracle_admm_equals_numpy_oracle(scheme, single):
    rng = np.random.default_rng(6)
    for shape, lz, mu, use_mask in GEOMS[:4]:
        mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=3.0 if use_mask else 0)
        x0 = 50 * rng.random(shape)
        wx, wloss, wz, wu = orc.admm(x0, 4, 5.0, 0.3, 3, scheme=scheme, return_state=True, single_reduction=single, **kw)
        gx, gloss, gz, gu = occ.admm(x0, 4, 5.0, 0.3, 3, scheme=scheme, return_state=True, single_reduction=single, **kw)
        np.testing.assert_allclose(gloss, wloss, rtol=1e-11)
        np.testing.assert_allclose(gx, wx, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gz, wz, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(gu, wu, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_single_reduction_cg_is_the_same_iteration(scheme):
    """The Chronopoulos-Gear recurrence and the textbook CG agree to rounding (same Krylov iterates)."""
    rng = np.random.default_rng(8)
    x0 = 50 * rng.random((4, 3, 7, 6))
    kw = dict(reg_z_over_reg=1.5, reg_time=0.7)
    ax, aloss = orc.admm(x0, 5, 5.0, 0.3, 4, scheme=scheme, **kw)
    bx, bloss = orc.admm(x0, 5, 5.0, 0.3, 4, scheme=scheme, single_reduction=True, **kw)
    np.testing.assert_allclose(bloss, aloss, rtol=1e-9)
    np.testing.assert_allclose(bx, ax, rtol=1e-8, atol=1e-8)


def test_c_oracle_numa_variant_is_the_same_iteration():
    rng = np.random.default_rng(12)
    x0 = (50 * rng.random((6, 3, 9, 10))).astype(np.float32)
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    ax, aloss = occ.chambolle_pock(x0, 7, 5.0, scheme="hybrid", **kw)
    bx, bloss, secs = occ.chambolle_pock(x0, 7, 5.0, scheme="hybrid", numa=True, **kw)
    assert np.array_equal(ax, bx) and secs > 0
    np.testing.assert_allclose(bloss, aloss, rtol=1e-12)          # OpenMP reductions: the summation order is not fixed

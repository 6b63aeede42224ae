"""CPU: the oracle's per-voxel weight extension (mask_static = float array of shape (Nz, M, Ny, Nx); the reference's
to-do README.md:258).  No reference code exists for it, so it is pinned by properties: it reduces to the per-pixel map
and to the reference's boolean mask (golden vectors), and D^T stays the exact adjoint of D."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCHEMES
from oracle import tv_oracle as orc


@pytest.mark.parametrize("scheme", SCHEMES)
def test_constant_volume_reproduces_the_reference_mask_golden(scheme):
    z = np.load(os.path.join(GOLDEN, "ops_%s.npz" % scheme))
    done = 0
    for name in z["case_names"]:
        name = str(name)
        mask = z[name + "/mask"]
        if mask.ndim == 0 or z[name + "/x"].dtype != np.float64:
            continue
        lz, mu, factor = z[name + "/params"]
        x, y = z[name + "/x"], z[name + "/y"]
        W = np.broadcast_to(np.where(mask, factor, 1.0), x.shape).copy()
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=W)
        np.testing.assert_allclose(orc.D(x, scheme, **kw), z[name + "/D"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(orc.D_T(y, scheme, **kw), z[name + "/DT"], rtol=1e-12, atol=1e-12)
        tv, G = orc.tv(x, scheme, **kw)
        np.testing.assert_allclose(tv, z[name + "/tv"], rtol=1e-12)
        np.testing.assert_allclose(G, z[name + "/G"], rtol=1e-11, atol=1e-12)
        done += 1
    assert done >= 1


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", [(4, 5, 6, 7), (1, 4, 5, 5), (3, 2, 4, 6)])
def test_adjointness_and_reduction_with_a_weight_volume(scheme, shape):
    rng = np.random.default_rng(2)
    W = rng.random(shape) * 3
    W[:, 1] = 0.0
    kw = dict(reg_z_over_reg=1.3, reg_time=0.7, mask_static=W)
    x = rng.standard_normal(shape)
    d = orc.D(x, scheme, **kw)
    y = rng.standard_normal(d.shape)
    assert abs(np.sum(d * y) - np.sum(x * orc.D_T(y, scheme, **kw))) < 1e-10
    # the time channel(s) of D carry sqrt(W) of their own voxel, nothing else changes
    d1 = orc.D(x, scheme, reg_z_over_reg=1.3, reg_time=0.7)
    nt = 2 if scheme == "hybrid" else 1
    np.testing.assert_allclose(d[:, -nt:], d1[:, -nt:] * np.sqrt(W)[:, None], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(d[:, :-nt], d1[:, :-nt])
    # a volume without z / t variation is the per-pixel map
    Wp = rng.random((1, 1) + shape[2:]) * 3
    a = orc.D_T(y, scheme, reg_z_over_reg=1.3, reg_time=0.7, mask_static=Wp)
    b = orc.D_T(y, scheme, reg_z_over_reg=1.3, reg_time=0.7, mask_static=np.broadcast_to(Wp, shape).copy())
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13)
    assert orc.time_weight_max(W, 0) == W.max()

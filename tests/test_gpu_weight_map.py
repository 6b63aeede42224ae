"""Per-pixel weight map of the time regularisation: ``mask_static=<float array W>`` (C-ABI field ``time_factor`` =
sqrt(W)).  This is the generalisation the reference's to-do list asks for (README.md:258, "replace mask_static,
factor_reg_static with a weight matrix"); it does not exist in the reference, so parity is pinned where it can be:
W = where(mask, factor, 1) must reproduce the reference's boolean-mask results (golden vectors), and general maps
follow the same formula in the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"


@pytest.fixture(scope="module")
def pytv():
    import pytv
    return pytv


@pytest.mark.parametrize("scheme", SCHEMES)
def test_two_valued_map_reproduces_the_reference_mask_golden(pytv, scheme):
    z = np.load(os.path.join(GOLDEN, "ops_%s.npz" % scheme))
    done = 0
    for name in z["case_names"]:
        name = str(name)
        mask = z[name + "/mask"]
        if mask.ndim == 0:
            continue
        lz, mu, factor = z[name + "/params"]
        x, y = z[name + "/x"], z[name + "/y"]
        W = np.where(mask, factor, 1.0)
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=W)
        tol = dict(rtol=1e-5, atol=1e-5) if x.dtype == np.float32 else dict(rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(getattr(pytv.tv_operators_GPU, "D_" + scheme)(x, **kw), z[name + "/D"], **tol)
        np.testing.assert_allclose(getattr(pytv.tv_operators_GPU, "D_T_" + scheme)(y, **kw), z[name + "/DT"], **tol)
        tv, G = getattr(pytv.tv_GPU, "tv_" + scheme)(x.copy(), **kw)
        np.testing.assert_allclose(float(tv), z[name + "/tv"], rtol=tol["rtol"])
        np.testing.assert_allclose(G, z[name + "/G"], **tol)
        done += 1
    assert done >= 2


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype,shape", [(np.float64, (4, 3, 9, 14)), (np.float32, (5, 4, 18, 132)), (np.float32, (6, 8, 12, 64))])
def test_general_weight_map_matches_oracle(pytv, scheme, dtype, shape):
    import torch
    from pytv import _native as nv
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(shape) * 10).astype(dtype)
    W = rng.random(shape[2:]) * 3.0
    W[2:4, 3:9] = 0.0                         # no time regularisation at all on a patch
    kw = dict(reg_z_over_reg=1.3, reg_time=0.8, mask_static=W)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == np.float32 else dict(rtol=1e-11, atol=1e-11)
    x64 = x.astype(np.float64)
    d = getattr(pytv.tv_operators_GPU, "D_" + scheme)(x, **kw)
    np.testing.assert_allclose(d, orc.D(x64, scheme, **kw), **tol)
    y = rng.standard_normal(d.shape).astype(dtype)
    np.testing.assert_allclose(getattr(pytv.tv_operators_GPU, "D_T_" + scheme)(y, **kw), orc.D_T(y.astype(np.float64), scheme, **kw), **tol)
    # TV + sub-gradient, with and without the norms (two-pass and one-pass kernels)
    tv_ref, G_ref = orc.tv(x64, scheme, **kw)
    for norms in (True, False):
        out = getattr(pytv.tv_GPU, "tv_" + scheme)(x.copy(), return_grad_norms=norms, **kw)
        np.testing.assert_allclose(float(out[0]), tv_ref, rtol=1e-6 if dtype == np.float32 else 1e-12)
        np.testing.assert_allclose(out[1], G_ref, **tol)
    # Chambolle-Pock, both paths where available
    x0 = torch.as_tensor(x * 5).cuda()
    ref_x, ref_loss = orc.chambolle_pock(x64 * 5, 8, 7.0, scheme=scheme, **kw)
    for fused in (False, None):
        cp = pytv.solvers.ChambollePock(x0, 7.0, scheme=scheme, fused=fused, **kw)
        loss = cp.run(8)
        np.testing.assert_allclose(loss, ref_loss, rtol=1e-5 if dtype == np.float32 else 1e-10)
        np.testing.assert_allclose(cp.result().cpu().numpy(), ref_x, rtol=1e-4, atol=1e-3 if dtype == np.float32 else 1e-8)
    # ADMM (normal operator, z/u update, D^T axpy all see the map)
    ad = pytv.solvers.ADMM(x0, 7.0, 0.1, n_cg=4, scheme=scheme, x_solver="cg", **kw)
    la = ad.run(3)
    _, lref = orc.admm(x64 * 5, 3, 7.0, 0.1, 4, scheme=scheme, single_reduction=True, **kw)
    np.testing.assert_allclose(la, lref, rtol=1e-6 if dtype == np.float32 else 1e-9)      # measured 3e-8: profiles/r3_admm_tolerances.txt
    # adjointness with the map
    g = nv.Geometry(shape, scheme, torch.float64 if dtype == np.float64 else torch.float32, "cuda", **kw)
    assert g.factor_dev is not None and g.mask_dev is None
    lhs = float((torch.as_tensor(d.astype(np.float64)) * torch.as_tensor(y.astype(np.float64))).sum())
    rhs = float((torch.as_tensor(x64) * torch.as_tensor(getattr(pytv.tv_operators_GPU, "D_T_" + scheme)(y, **kw).astype(np.float64))).sum())
    assert abs(lhs - rhs) <= (1e-4 if dtype == np.float32 else 1e-10) * abs(lhs)


def test_negative_weights_are_rejected(pytv):
    with pytest.raises(ValueError):
        pytv.tv_operators_GPU.D_hybrid(np.zeros((2, 3, 8, 8), np.float32), reg_time=1.0, mask_static=-np.ones((8, 8)))

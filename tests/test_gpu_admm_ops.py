"""The C-ABI entry points round 2 added for the ADMM x-update (include/pytv4d.h): tv_normal_op2 (streaming normal operator
with an optional right-hand side and two dot products), tv_cg_update (one step of the single-reduction CG), tv_admm_tu (the
z / u update that stores t = z - u).  Each against the oracle / a NumPy restatement, fp32 on the streaming kernels
(k_normal_stream, k_normal_stream_cen: all four schemes, M <= 8 and time windows, z-chunk edges, slabs with two-plane halos
bit-equal to the unsharded call) and fp64 / small frames on the composed fallback."""
import os

import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu

os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"      # small test volumes take the one-sweep Chambolle-Pock path too

SHAPES = [(7, 3, 9, 256), (6, 2, 5, 132), (9, 8, 6, 192), (3, 16, 5, 128), (4, 12, 7, 64), (1, 1, 33, 68), (1, 4, 8, 64), (8, 5, 3, 64),
          (1, 20, 2, 72), (11, 1, 1, 260), (5, 3, 8, 12)]


@pytest.fixture(scope="module")
def nvlib():
    import pytv  # noqa: F401
    from pytv import _native as nv
    return nv


def _A(x64, scheme, rho, kw):
    return x64 + rho * orc.D_T(orc.D(x64, scheme, **kw), scheme, **kw)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("zchunk", [0, 3])
def test_normal_op2_matches_oracle(nvlib, scheme, zchunk, tvopt):
    import torch
    nv, lib = nvlib, nvlib.lib()
    tvopt("TV_ZCHUNK", zchunk)
    rng = np.random.default_rng(5 + zchunk)
    for shape in SHAPES:
        if scheme == "central" and (shape[0] == 2 or shape[1] == 2):
            continue
        for dtype in ((np.float32, np.float64) if shape[-1] <= 72 else (np.float32,)):
            use_mask = shape[-1] == 192
            kw = dict(reg_z_over_reg=1.3, reg_time=0.5)
            if use_mask:
                kw.update(mask_static=rng.random(shape[2:]) < 0.4, factor_reg_static=2.5)
            x = torch.as_tensor((rng.standard_normal(shape) * 10).astype(dtype)).cuda()
            b = torch.as_tensor((rng.standard_normal(shape) * 10).astype(dtype)).cuda()
            g = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
            st, ws = nv.current_stream(x.device), g.workspace()
            x64, b64 = x.double().cpu().numpy(), b.double().cpu().numpy()
            want = _A(x64, scheme, 0.3, kw)
            tol = dict(rtol=1e-5, atol=2e-3) if dtype == np.float32 else dict(rtol=1e-11, atol=1e-10)
            dtol = 1e-5 if dtype == np.float32 else 1e-12
            for mode in ("plain", "residual"):
                out, out2 = torch.empty_like(x), torch.empty_like(x)
                dots = torch.zeros(2, dtype=torch.float64, device="cuda")
                nv.check(lib.tv_normal_op2(g.ref, nv.ptr(x), None, None, 0.3, nv.ptr(b) if mode == "residual" else None, nv.ptr(out),
                                           nv.ptr(out2) if mode == "residual" else None, dots.data_ptr(), nv.ptr(ws), st))
                w = want if mode == "plain" else b64 - want
                np.testing.assert_allclose(out.cpu().numpy(), w, err_msg="%s %s %s" % (scheme, shape, mode), **tol)
                d0 = float(np.sum(x64 * want)) if mode == "plain" else float(np.sum(w * w))
                assert abs(dots[0].item() - d0) <= dtol * abs(d0) and abs(dots[1].item() - float(np.sum(x64 * x64))) <= dtol * float(np.sum(x64 * x64))
                if mode == "residual":
                    assert torch.equal(out, out2)
            # the plain entry point gives the same vector and the first dot product
            o1, d1 = torch.empty_like(x), torch.zeros(1, dtype=torch.float64, device="cuda")
            nv.check(lib.tv_normal_op(g.ref, nv.ptr(x), None, None, 0.3, nv.ptr(o1), d1.data_ptr(), nv.ptr(ws), st))
            np.testing.assert_allclose(o1.cpu().numpy(), want, **tol)
            assert abs(d1.item() - float(np.sum(x64 * want))) <= dtol * abs(float(np.sum(x64 * want)))


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", [(9, 3, 6, 128), (8, 12, 5, 64)])
def test_normal_op2_on_slabs_equals_unsharded(nvlib, scheme, shape, tvopt):
    """z-slabs with TWO-plane halos (zero planes stand in where the volume ends) == the unsharded call, bit for bit."""
    import torch
    nv, lib = nvlib, nvlib.lib()
    tvopt("TV_ZCHUNK", 2)
    rng = np.random.default_rng(8)
    kw = dict(reg_z_over_reg=1.2, reg_time=0.9)
    x = torch.as_tensor((rng.standard_normal(shape) * 10).astype(np.float32)).cuda()
    b = torch.as_tensor((rng.standard_normal(shape) * 10).astype(np.float32)).cuda()
    gF = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
    st = nv.current_stream(x.device)
    oF, dF = torch.empty_like(x), torch.zeros(2, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_normal_op2(gF.ref, nv.ptr(x), None, None, 0.3, nv.ptr(b), nv.ptr(oF), None, dF.data_ptr(), nv.ptr(gF.workspace()), st))
    nz = shape[0]
    tot = np.zeros(2)
    zero2 = torch.zeros((2,) + shape[1:], device="cuda")
    for a, e in ((0, 2), (2, 5), (5, nz - 1), (nz - 1, nz)):
        gS = nv.Geometry((e - a,) + shape[1:], scheme, x.dtype, x.device, nz_global=nz, z0=a, **kw)
        xp = None if a == 0 else (x[a - 2:a] if a >= 2 else torch.cat([zero2[0:1], x[0:1]]))
        xn = None if e == nz else (x[e:e + 2] if e + 2 <= nz else torch.cat([x[e:e + 1], zero2[0:1]]))
        oS, dS = torch.empty_like(x[a:e]), torch.zeros(2, dtype=torch.float64, device="cuda")
        nv.check(lib.tv_normal_op2(gS.ref, nv.ptr(x[a:e]), nv.ptr(xp), nv.ptr(xn), 0.3, nv.ptr(b[a:e]), nv.ptr(oS), None, dS.data_ptr(),
                                   nv.ptr(gS.workspace()), st))
        assert torch.equal(oS, oF[a:e]), (scheme, a, e)
        tot += dS.cpu().numpy()
    np.testing.assert_allclose(tot, dF.cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_update_follows_the_chronopoulos_gear_recurrence(nvlib, dtype):
    import torch
    nv, lib = nvlib, nvlib.lib()
    shape = (3, 2, 7, 24)
    rng = np.random.default_rng(3)
    g = nv.Geometry(shape, "upwind", torch.float32 if dtype == np.float32 else torch.float64, "cuda")
    v = {k: rng.standard_normal(shape).astype(dtype) for k in "xrdsw"}
    t = {k: torch.as_tensor(a.copy()).cuda() for k, a in v.items()}
    x0 = torch.as_tensor(rng.standard_normal(shape).astype(dtype)).cuda()
    st, ws = nv.current_stream(x0.device), g.workspace()
    sc = torch.tensor([3.0, 5.0, 0.0, 0.0], dtype=torch.float64, device="cuda")        # gamma, delta, first step
    fid = torch.zeros(1, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_cg_update(g.ref, nv.ptr(t["x"]), nv.ptr(t["r"]), nv.ptr(t["d"]), nv.ptr(t["s"]), nv.ptr(t["w"]), sc.data_ptr(), None, None,
                              nv.ptr(ws), st))
    alpha = dtype(3.0 / 5.0)
    d1, s1 = v["r"], v["w"]                                        # beta = 0: whatever d and s held is ignored
    x1, r1 = v["x"] + alpha * d1, v["r"] - alpha * s1
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == np.float32 else dict(rtol=1e-14, atol=1e-14)
    for k, wv in (("d", d1), ("s", s1), ("x", x1), ("r", r1)):
        np.testing.assert_allclose(t[k].cpu().numpy(), wv, **tol)
    assert sc.tolist()[2:] == [3.0, 0.6]
    # second step: new gamma / delta, beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old)
    w2 = rng.standard_normal(shape).astype(dtype)
    t["w"].copy_(torch.as_tensor(w2))
    sc[0], sc[1] = 2.0, 4.0
    nv.check(lib.tv_cg_update(g.ref, nv.ptr(t["x"]), nv.ptr(t["r"]), nv.ptr(t["d"]), nv.ptr(t["s"]), nv.ptr(t["w"]), sc.data_ptr(), nv.ptr(x0),
                              fid.data_ptr(), nv.ptr(ws), st))
    beta = 2.0 / 3.0
    al2 = 2.0 / (4.0 - beta * 2.0 / 0.6)
    d2, s2 = r1 + dtype(beta) * d1, w2 + dtype(beta) * s1
    x2, r2 = x1 + dtype(al2) * d2, r1 - dtype(al2) * s2
    tol2 = dict(rtol=1e-5, atol=1e-5) if dtype == np.float32 else dict(rtol=1e-13, atol=1e-13)
    for k, wv in (("d", d2), ("s", s2), ("x", x2), ("r", r2)):
        np.testing.assert_allclose(t[k].cpu().numpy(), wv, **tol2)
    assert abs(sc[3].item() - al2) < 1e-12 and sc[2].item() == 2.0
    want_fid = 0.5 * float(np.sum((x2.astype(np.float64) - x0.double().cpu().numpy()) ** 2))
    assert abs(fid.item() - want_fid) <= (1e-5 if dtype == np.float32 else 1e-12) * want_fid


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape,dtype", [((5, 3, 8, 132), np.float32), ((4, 2, 6, 10), np.float64), ((3, 16, 5, 128), np.float32)])
def test_admm_tu_is_admm_zu_with_t_equal_z_minus_u(nvlib, scheme, shape, dtype):
    import torch
    nv, lib = nvlib, nvlib.lib()
    if scheme == "central" and shape[1] == 2:
        pytest.skip("two-point time axis with central: covered by the z/u tests")
    rng = np.random.default_rng(4)
    kw = dict(reg_z_over_reg=1.0, reg_time=0.7)
    x = torch.as_tensor((rng.standard_normal(shape) * 10).astype(dtype)).cuda()
    g = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
    u0 = torch.as_tensor((rng.standard_normal(g.grad_shape) * 5).astype(dtype)).cuda()
    st, ws = nv.current_stream(x.device), g.workspace()
    z, u = torch.empty_like(u0), u0.clone()
    tv1 = torch.zeros(1, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_admm_zu(g.ref, nv.ptr(x), None, None, nv.ptr(z), nv.ptr(u), 3.0, tv1.data_ptr(), nv.ptr(ws), st))
    t, u2 = torch.empty_like(u0), u0.clone()
    tv2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_admm_tu(g.ref, nv.ptr(x), None, None, nv.ptr(t), nv.ptr(u2), 3.0, tv2.data_ptr(), nv.ptr(ws), st))
    assert torch.equal(u, u2) and tv1.item() == tv2.item()
    assert torch.allclose(t, z - u, rtol=1e-6, atol=1e-6)        # z - (v - z): the compiler may contract it differently
    v = orc.D(x.double().cpu().numpy(), scheme, **kw) + u0.double().cpu().numpy()
    wz = orc.group_soft_threshold(v, 3.0)
    tol = dict(rtol=1e-5, atol=1e-4) if dtype == np.float32 else dict(rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(t.cpu().numpy(), 2 * wz - v, **tol)


@pytest.mark.parametrize("single", [True, False])
@pytest.mark.parametrize("scheme", ["hybrid", "central"])
def test_admm_graph_replay_equals_eager(nvlib, scheme, single):
    """Small problems replay blocks of outer iterations from a hipGraph (no host round trip inside an outer iteration): same
    trajectory as the eager loop, bit for bit."""
    import torch
    import pytv
    rng = np.random.default_rng(6)
    x0 = torch.as_tensor((rng.random((1, 1, 96, 128)) * 100).astype(np.float32)).cuda()
    a = pytv.solvers.ADMM(x0, 20.0, 0.1, n_cg=3, scheme=scheme, single_reduction=single, x_solver="cg")
    b = pytv.solvers.ADMM(x0, 20.0, 0.1, n_cg=3, scheme=scheme, single_reduction=single, x_solver="cg")
    la, lb = a.run(19, graph=True), b.run(19, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result())
    assert la[-1] < la[0]

"""One-sweep dual side of ADMM (round 3; include/pytv4d.h: tv_admm_fused + tv_admm_fixup): the z / u update and the residual
r = [x0 + rho D^T (z - u)] - (I + rho D^T D) x of the next x-solve from ONE pass over u.  Against the oracle's ADMM (the
reference ships none: README.md:26,135 name it only -- parity is pinned op by op against the reference's D / D^T, SURVEY 8a-3
row a9), against a NumPy restatement of the two calls, and against the kernel trio it replaces."""
import os

import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu

os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"

# (Nz, M, Ny, Nx): one frame, several frames, more than 8 frames (time windows), more rows than a wave tile, more columns than a
# block tile (256 fp32 / 128 fp64 columns), ragged in everything
SHAPES = [(1, 1, 24, 64), (5, 3, 16, 64), (4, 10, 9, 128), (3, 2, 19, 324), (9, 8, 6, 192)]


@pytest.fixture(scope="module")
def nvlib():
    import pytv  # noqa: F401
    from pytv import _native as nv
    return nv


def _shrink(v, thresh):
    nv = np.sqrt(np.sum(v * v, axis=1, keepdims=True))
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(nv > 0, np.maximum(0.0, 1.0 - thresh / nv), 0.0)
    return v * sc


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("zchunk", [0, 2])
def test_admm_fused_calls_match_numpy(nvlib, scheme, shape, zchunk, tvopt):
    """tv_admm_fused + tv_admm_fixup on random x, u against NumPy over the oracle's D / D^T: u', r, TV, <r, r>, and t' where
    it is stored in full."""
    import torch
    nv, lib = nvlib, nvlib.lib()
    tvopt("TV_ZCHUNK", zchunk)
    rng = np.random.default_rng(11)
    kw = dict(reg_z_over_reg=1.3, reg_time=0.6)
    thresh, rho = 0.7, 0.15
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-5)):
        x = (rng.random(shape) * 4).astype(dtype)
        x0 = (x + rng.random(shape)).astype(dtype)
        dx = orc.D(x.astype(np.float64), scheme, **kw)
        u = (rng.standard_normal(dx.shape)).astype(dtype)
        v = dx + u.astype(np.float64)
        z = _shrink(v, thresh)
        un = v - z
        t = z - un
        r = (x0.astype(np.float64) - x) + rho * orc.D_T(t - dx, scheme, **kw)
        tv = np.sum(np.sqrt(np.sum(dx * dx, axis=1)))
        for full in (1, 0):
            g = nv.Geometry(shape, scheme, torch.as_tensor(x).dtype, torch.device("cuda", 0), **kw)
            assert lib.tv_cp_fused_supported(g.ref) == 1
            st, ws = nv.current_stream(torch.device("cuda", 0)), g.workspace()
            xd, x0d, ud = torch.as_tensor(x).cuda(), torch.as_tensor(x0).cuda(), torch.as_tensor(u).cuda()
            td = torch.full_like(ud, 7.0)
            rd = torch.empty_like(xd)
            sc = torch.zeros(3, dtype=torch.float64, device="cuda")
            nv.check(lib.tv_admm_fused(g.ref, nv.ptr(xd), None, None, nv.ptr(ud), nv.ptr(td), nv.ptr(x0d), nv.ptr(rd), thresh, rho, full,
                                       0, -1, sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(ws), st))
            nv.check(lib.tv_admm_fixup(g.ref, nv.ptr(td), None, None, nv.ptr(rd), rho, 0, -1, sc[2:3].data_ptr(), nv.ptr(ws), st))
            scale = max(1.0, np.abs(r).max())
            np.testing.assert_allclose(ud.cpu().numpy(), un, rtol=0, atol=tol * 10, err_msg="u %s %s" % (dtype, full))
            np.testing.assert_allclose(rd.cpu().numpy(), r, rtol=0, atol=tol * 10 * scale, err_msg="r %s %s" % (dtype, full))
            s = sc.cpu().numpy()
            np.testing.assert_allclose(s[0], tv, rtol=max(tol, 1e-12) * 10)
            np.testing.assert_allclose(s[1] + s[2], np.sum(r * r), rtol=tol * 100)
            if full:
                np.testing.assert_allclose(td.cpu().numpy(), t - dx, rtol=0, atol=tol * 10)
            else:
                assert (td == 7.0).float().mean().item() > 0.3          # most samples are never written
            # tv_admm_sweep (round 5): the same sweep reading u from one array and writing another -- bit-identical u', r, t', scalars,
            # the array it read untouched (what lets solvers.ADMM rebuild z = shrink(D x + u_in) on demand)
            u_in, u_out = torch.as_tensor(u).cuda(), torch.full_like(ud, 3.0)
            td2, rd2, sc2 = torch.full_like(ud, 7.0), torch.empty_like(xd), torch.zeros(3, dtype=torch.float64, device="cuda")
            nv.check(lib.tv_admm_sweep(g.ref, nv.ptr(xd), None, None, nv.ptr(u_in), nv.ptr(u_out), nv.ptr(td2), nv.ptr(x0d), nv.ptr(rd2), thresh,
                                       rho, full, 0, -1, sc2[0:1].data_ptr(), sc2[1:2].data_ptr(), nv.ptr(ws), st))
            nv.check(lib.tv_admm_fixup(g.ref, nv.ptr(td2), None, None, nv.ptr(rd2), rho, 0, -1, sc2[2:3].data_ptr(), nv.ptr(ws), st))
            assert torch.equal(u_out, ud) and torch.equal(rd2, rd) and torch.equal(td2, td) and torch.equal(sc2, sc)
            assert torch.equal(u_in, torch.as_tensor(u).cuda())
        # bit 1 of full_store: the second partial is |x - x0|^2 over all sites (what the Chebyshev x-solve asks for)
        ud = torch.as_tensor(u).cuda()
        nv.check(lib.tv_admm_fused(g.ref, nv.ptr(xd), None, None, nv.ptr(ud), nv.ptr(td), nv.ptr(x0d), nv.ptr(rd), thresh, rho, 2,
                                   0, -1, sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(ws), st))
        np.testing.assert_allclose(sc[1].item(), np.sum((x.astype(np.float64) - x0.astype(np.float64)) ** 2), rtol=max(tol, 1e-12) * 10)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("reg,rho", [(25.0, 0.05), (4.0, 0.1)])      # threshold reg / rho = 500 (z stays 0) and 40 (z active)
@pytest.mark.parametrize("shape,lz,mu", [((1, 1, 24, 64), 1.0, 0.0), ((5, 3, 16, 64), 1.5, 0.5), ((3, 10, 9, 128), 1.0, 0.7)])
def test_admm_fused_matches_oracle(scheme, shape, lz, mu, reg, rho):
    """fp32 bounds: ~10 x the measured deviation from the fp64 oracle (profiles/r3_admm_tolerances.txt)."""
    import torch
    import pytv
    rng = np.random.default_rng(6)
    for dtype, rtol, atol in ((np.float64, 1e-9, 1e-8), (np.float32, 2e-6, 2e-4)):
        x0 = (rng.random(shape) * 100).astype(dtype)
        wx, wloss, wz, wu = orc.admm(x0.astype(np.float64), 6, reg, rho, 5, scheme=scheme, reg_z_over_reg=lz, reg_time=mu,
                                     single_reduction=True, return_state=True)
        if rho == 0.1:
            assert np.abs(wz).max() > 1.0              # the shrinkage branch is exercised
        for keep_z in (True, False):
            ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), reg, rho, n_cg=5, scheme=scheme, reg_z_over_reg=lz, reg_time=mu,
                                   keep_z=keep_z, x_solver="cg")
            assert ad.fused
            loss = ad.run(6)
            np.testing.assert_allclose(loss, wloss, rtol=rtol / 2, err_msg="%s %s" % (scheme, shape))
            np.testing.assert_allclose(ad.result().cpu().numpy(), wx, rtol=rtol, atol=atol)
            np.testing.assert_allclose(ad.u.cpu().numpy(), wu, rtol=rtol * 10, atol=atol * 3)
            if keep_z:
                np.testing.assert_allclose(ad.z.cpu().numpy(), wz, rtol=rtol * 10, atol=atol * 3)
            else:
                with pytest.raises(RuntimeError):
                    ad.z


@pytest.mark.parametrize("scheme", SCHEMES)
def test_admm_fused_equals_kernel_trio(scheme):
    """The one-sweep path against tv_admm_tu + tv_DT_axpy + tv_normal_op2 on a volume with many tiles, chunks and a time-window
    seam: the same iteration up to the rounding of r (formed as one sum instead of two); sparse == full storage of t' bit for bit."""
    import torch
    import pytv
    rng = np.random.default_rng(8)
    x0 = torch.as_tensor((rng.random((10, 12, 40, 320)) * 100).astype(np.float32)).cuda()
    kw = dict(n_cg=4, scheme=scheme, reg_time=0.8, x_solver="cg")
    a = pytv.solvers.ADMM(x0, 4.0, 0.1, fused=True, **kw)             # threshold 40 against differences of +-100: z is active
    b = pytv.solvers.ADMM(x0, 4.0, 0.1, fused=False, **kw)
    c = pytv.solvers.ADMM(x0, 4.0, 0.1, fused=True, keep_z=True, **kw)
    la, lb, lc = a.run(5), b.run(5), c.run(5)
    assert np.array_equal(la, lc)
    assert torch.equal(a.result(), c.result()) and torch.equal(a.u, c.u)
    np.testing.assert_allclose(la, lb, rtol=2e-6)
    np.testing.assert_allclose(a.result().cpu().numpy(), b.result().cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(c.z.cpu().numpy(), b.z.cpu().numpy(), rtol=0, atol=2e-3)
    assert c.z.abs().max().item() > 1.0
    assert la[-1] < la[0]


# ------------------------------------------------------------------------------------------------
# Chebyshev x-solve (tv_cheb_step, tv_axpby; solvers.ADMM(x_solver="chebyshev"))
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", SHAPES + [(5, 3, 8, 12), (7, 2, 6, 70)])          # the last two: the composed fallback (ragged Nx)
def test_cheb_step_matches_numpy(nvlib, scheme, shape):
    """out = add + x + alpha (b - A x) + beta (x - y) and its two reductions against NumPy over the oracle's D / D^T; the streaming
    kernels (fp32 all schemes, fp64 one-sided schemes) and the composition of tv_normal_op with one flat pass (everything else)."""
    import torch
    nv, lib = nvlib, nvlib.lib()
    rng = np.random.default_rng(21)
    kw = dict(reg_z_over_reg=1.3, reg_time=0.6)
    rho, alpha, beta = 0.15, 0.8, 0.3
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-5)):
        x, y, b, add, ref = [(rng.standard_normal(shape) * 3).astype(dtype) for _ in range(5)]
        x64 = x.astype(np.float64)
        ax = x64 + rho * orc.D_T(orc.D(x64, scheme, **kw), scheme, **kw)
        res = b - ax
        g = nv.Geometry(shape, scheme, torch.as_tensor(x).dtype, torch.device("cuda", 0), **kw)
        st, ws = nv.current_stream(torch.device("cuda", 0)), g.workspace()
        xd, yd, bd, addd, refd = [torch.as_tensor(v).cuda() for v in (x, y, b, add, ref)]
        for use_y, yscale, use_add, use_ref in ((True, 0.0, True, True), (False, 0.0, False, False), (True, 0.0, False, False),
                                                (False, 0.4, False, True)):          # y missing: y = yscale * b
            want = x64 + alpha * res + beta * (x64 - (y if use_y else yscale * b.astype(np.float64))) + (add if use_add else 0.0)
            od = torch.empty_like(xd)
            sc = torch.zeros(2, dtype=torch.float64, device="cuda")
            nv.check(lib.tv_cheb_step(g.ref, nv.ptr(xd), None, None, rho, nv.ptr(bd), nv.ptr(yd) if use_y else None, yscale,
                                      nv.ptr(addd) if use_add else None, nv.ptr(refd) if use_ref else None, alpha, beta, nv.ptr(od),
                                      sc.data_ptr(), nv.ptr(ws), st))
            scale = max(1.0, np.abs(want).max())
            np.testing.assert_allclose(od.cpu().numpy(), want, rtol=0, atol=tol * 10 * scale)
            s_ = sc.cpu().numpy()
            np.testing.assert_allclose(s_[0], np.sum(res * res), rtol=tol * 100)
            np.testing.assert_allclose(s_[1], np.sum((want - ref) ** 2) if use_ref else np.sum(x64 * x64), rtol=tol * 100)
            # dots = NULL (round 5): the same output bit for bit, no dot products (refused together with ref, which exists for one of them)
            od2 = torch.full_like(xd, float("nan"))
            rc = lib.tv_cheb_step(g.ref, nv.ptr(xd), None, None, rho, nv.ptr(bd), nv.ptr(yd) if use_y else None, yscale,
                                  nv.ptr(addd) if use_add else None, nv.ptr(refd) if use_ref else None, alpha, beta, nv.ptr(od2), None, nv.ptr(ws), st)
            if use_ref:
                assert rc != 0 and b"ref" in lib.tv_last_error()
            else:
                nv.check(rc)
                assert torch.equal(od2, od)
        # x may be b itself (the fused first two steps of the solver)
        od = torch.empty_like(xd)
        nv.check(lib.tv_cheb_step(g.ref, nv.ptr(bd), None, None, rho, nv.ptr(bd), None, 0.0, None, None, alpha, beta, nv.ptr(od),
                                  sc.data_ptr(), nv.ptr(ws), st))
        b64 = b.astype(np.float64)
        ab = b64 + rho * orc.D_T(orc.D(b64, scheme, **kw), scheme, **kw)
        want = b64 + alpha * (b64 - ab) + beta * b64
        np.testing.assert_allclose(od.cpu().numpy(), want, rtol=0, atol=tol * 10 * max(1.0, np.abs(want).max()))
        # tv_axpby
        od = torch.empty_like(xd)
        sc = torch.zeros(1, dtype=torch.float64, device="cuda")
        nv.check(lib.tv_axpby(g.ref, 0.7, nv.ptr(xd), -1.3, nv.ptr(yd), nv.ptr(refd), nv.ptr(od), sc.data_ptr(), nv.ptr(ws), st))
        want = 0.7 * x64 - 1.3 * y
        np.testing.assert_allclose(od.cpu().numpy(), want, rtol=0, atol=tol * 10)
        np.testing.assert_allclose(sc.item(), np.sum((want - ref) ** 2), rtol=tol * 100)
        nv.check(lib.tv_axpby(g.ref, 0.5, nv.ptr(xd), 0.0, None, None, nv.ptr(od), None, None, st))
        np.testing.assert_allclose(od.cpu().numpy(), 0.5 * x64, rtol=0, atol=tol)
        # out = NULL with a reference: the distance alone, the same number, nothing stored (round 4)
        sc2 = torch.zeros(1, dtype=torch.float64, device="cuda")
        nv.check(lib.tv_axpby(g.ref, 0.7, nv.ptr(xd), -1.3, nv.ptr(yd), nv.ptr(refd), None, sc2.data_ptr(), nv.ptr(ws), st))
        assert sc2.item() == sc.item()
        assert lib.tv_axpby(g.ref, 0.7, nv.ptr(xd), -1.3, nv.ptr(yd), None, None, None, None, st) != 0      # neither an output nor a distance: refused


def test_chebyshev_coefficients_and_bound():
    """The recurrence of pytv.solvers restates the oracle's; the bound L really bounds D^T D (power iteration on the oracle)."""
    import pytv
    for lmax in (1.0, 1.8, 17.0):
        a = pytv.solvers.chebyshev_coefficients(lmax, 6)
        b = orc.chebyshev_coefficients(lmax, 6)
        np.testing.assert_allclose(a, b, rtol=1e-15)
    rng = np.random.default_rng(3)
    shape, kw = (4, 3, 10, 12), dict(reg_z_over_reg=1.5, reg_time=0.7)
    for scheme in SCHEMES:
        v = rng.standard_normal(shape)
        for _ in range(150):
            w = orc.D_T(orc.D(v, scheme, **kw), scheme, **kw)
            lam = np.linalg.norm(w) / np.linalg.norm(v)
            v = w / np.linalg.norm(w)
        L = pytv.solvers.normal_spectral_bound(scheme, shape[0], shape[1], kw["reg_z_over_reg"], kw["reg_time"])
        assert L == orc.normal_spectral_bound(scheme, shape, **kw)
        assert lam <= L and lam > 0.6 * L


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape,lz,mu", [((1, 1, 24, 64), 1.0, 0.0), ((5, 3, 16, 64), 1.5, 0.5), ((3, 10, 9, 128), 1.0, 0.7), ((5, 3, 8, 12), 1.5, 0.5)])
def test_admm_chebyshev_matches_oracle(scheme, shape, lz, mu):
    """ADMM with the Chebyshev x-solve (one-sweep dual side where the geometry allows, the kernel trio otherwise) against the
    oracle's restatement; and it reaches the objective of the CG variant (same number of steps) to 1e-5."""
    import torch
    import pytv
    rng = np.random.default_rng(6)
    reg, rho = 4.0, 0.1
    for dtype, rtol, atol in ((np.float64, 1e-9, 1e-8), (np.float32, 2e-6, 2e-4)):
        x0 = (rng.random(shape) * 100).astype(dtype)
        for n_cg in (1, 2, 5):
            wx, wloss, wz, wu = orc.admm(x0.astype(np.float64), 6, reg, rho, n_cg, scheme=scheme, reg_z_over_reg=lz, reg_time=mu,
                                         x_solver="chebyshev", return_state=True)
            for fused in ((True, False) if shape[-1] >= 64 else (False,)):
                ad = pytv.solvers.ADMM(torch.as_tensor(x0).cuda(), reg, rho, n_cg=n_cg, scheme=scheme, reg_z_over_reg=lz, reg_time=mu,
                                       x_solver="chebyshev", fused=fused, keep_z=True)
                loss = ad.run(6)
                np.testing.assert_allclose(loss, wloss, rtol=rtol / 2, err_msg="%s %s n_cg=%d fused=%s" % (scheme, shape, n_cg, fused))
                np.testing.assert_allclose(ad.result().cpu().numpy(), wx, rtol=rtol, atol=atol)
                np.testing.assert_allclose(ad.z.cpu().numpy(), wz, rtol=rtol * 10, atol=atol * 3)
                np.testing.assert_allclose(ad.u.cpu().numpy(), wu, rtol=rtol * 10, atol=atol * 3)
        _, closs = orc.admm(x0.astype(np.float64), 6, reg, rho, 5, scheme=scheme, reg_z_over_reg=lz, reg_time=mu, single_reduction=True)
        np.testing.assert_allclose(wloss[-1], closs[-1], rtol=1e-5)


@pytest.mark.parametrize("scheme", ["hybrid", "central"])
def test_admm_chebyshev_graph_replay_equals_eager(scheme):
    import torch
    import pytv
    rng = np.random.default_rng(6)
    x0 = torch.as_tensor((rng.random((1, 1, 96, 128)) * 100).astype(np.float32)).cuda()
    a = pytv.solvers.ADMM(x0, 4.0, 0.1, n_cg=3, scheme=scheme, x_solver="chebyshev")
    b = pytv.solvers.ADMM(x0, 4.0, 0.1, n_cg=3, scheme=scheme, x_solver="chebyshev")
    la, lb = a.run(19, graph=True), b.run(19, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result())
    assert la[-1] < la[0]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_admm_sweep_and_cheb_step_stay_inside_their_arrays(nvlib, scheme, dtype):
    """Every array of tv_admm_fused / tv_admm_fixup / tv_cheb_step sits between NaN guard bands: nothing outside is written, and
    (inputs between NaNs) nothing outside is read into a result."""
    import torch
    nv, lib = nvlib, nvlib.lib()
    shape = (4, 10, 19, 132)                     # ragged in rows, two block tiles in fp64, a time-window seam
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    g = nv.Geometry(shape, scheme, tdt, torch.device("cuda", 0), reg_z_over_reg=1.2, reg_time=0.8)
    pad = 4100
    rng = np.random.default_rng(3)

    def guarded(shp, fill=None):
        n = int(np.prod(shp))
        buf = torch.full((n + 2 * pad,), float("nan"), device="cuda", dtype=tdt)
        v = buf[pad:pad + n].view(shp)
        v.copy_(torch.as_tensor(rng.standard_normal(shp).astype(dtype)).cuda() if fill is None else torch.full(shp, fill, dtype=tdt, device="cuda"))
        return buf, v

    def bands_intact(buf):
        return bool(torch.isnan(buf[:pad]).all() and torch.isnan(buf[-pad:]).all())

    st, ws = nv.current_stream(torch.device("cuda", 0)), g.workspace()
    (xb, x), (x0b, x0), (ub, u), (tb, t), (rb, r) = guarded(shape), guarded(shape), guarded(g.grad_shape), guarded(g.grad_shape, 0.0), guarded(shape, 0.0)
    sc = torch.zeros(3, dtype=torch.float64, device="cuda")
    for full in (0, 1):
        nv.check(lib.tv_admm_fused(g.ref, nv.ptr(x), None, None, nv.ptr(u), nv.ptr(t), nv.ptr(x0), nv.ptr(r), 0.7, 0.15, full, 0, -1,
                                   sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(ws), st))
        nv.check(lib.tv_admm_fixup(g.ref, nv.ptr(t), None, None, nv.ptr(r), 0.15, 0, -1, sc[2:3].data_ptr(), nv.ptr(ws), st))
        torch.cuda.synchronize()
        assert all(bands_intact(b) for b in (xb, x0b, ub, tb, rb))
        assert bool(torch.isfinite(u).all() and torch.isfinite(t).all() and torch.isfinite(r).all()) and bool(torch.isfinite(sc).all())
    (yb, y), (ab, a), (ob, o) = guarded(shape), guarded(shape), guarded(shape, 0.0)
    dots = torch.zeros(2, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_cheb_step(g.ref, nv.ptr(x), None, None, 0.15, nv.ptr(r), nv.ptr(y), 0.0, nv.ptr(a), nv.ptr(x0), 0.8, 0.3, nv.ptr(o),
                              dots.data_ptr(), nv.ptr(ws), st))
    torch.cuda.synchronize()
    assert all(bands_intact(b) for b in (xb, rb, yb, ab, x0b, ob))
    assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(dots).all())


@pytest.mark.parametrize("fail_at", [1, 2, 3])
def test_admm_chebyshev_failed_graph_capture_leaves_the_state_intact(monkeypatch, fail_at):
    """round-3 advice: the Chebyshev x-solve rebinds x / d / Ad / b inside every step; a hipGraph capture that dies after an odd
    number of steps must not leave them pointing at buffers whose kernels never ran.  The capture is made to fail after
    `fail_at` steps; the run must then continue eagerly and give exactly the eager result."""
    import torch
    import pytv
    rng = np.random.default_rng(7)
    x0 = torch.as_tensor((rng.random((1, 1, 96, 128)) * 100).astype(np.float32)).cuda()
    a = pytv.solvers.ADMM(x0, 4.0, 0.1, n_cg=3, scheme="hybrid", x_solver="chebyshev")
    b = pytv.solvers.ADMM(x0, 4.0, 0.1, n_cg=3, scheme="hybrid", x_solver="chebyshev")
    real_step = pytv.solvers.ADMM.step
    state = {"capturing": False, "n": 0}
    real_graph = torch.cuda.graph

    class _FailingCapture(real_graph):
        def __enter__(self):
            state["capturing"], state["n"] = True, 0
            return super().__enter__()

        def __exit__(self, *exc):
            state["capturing"] = False
            return super().__exit__(*exc)

    def step(self, out):
        if state["capturing"] and self is a:
            if state["n"] == fail_at:
                raise RuntimeError("injected capture failure")
            state["n"] += 1
        return real_step(self, out)

    monkeypatch.setattr(torch.cuda, "graph", _FailingCapture)
    monkeypatch.setattr(pytv.solvers.ADMM, "step", step)
    la = a.run(13, graph=True)
    monkeypatch.setattr(torch.cuda, "graph", real_graph)
    lb = b.run(13, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result())

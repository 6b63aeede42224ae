"""Oracle parity of the HIP path AT the BASELINE.json configurations (full sizes, production dispatch):

  configs[1]  3-D (256, 1, 512, 512) hybrid Chambolle-Pock                 whole volume against the C / OpenMP oracle (fp64)
  configs[2]  4-D (128, 8, 512, 512) hybrid Chambolle-Pock, both paths     (y, x)-crops holding the FULL z and t extent
                                                                           against the C / OpenMP oracle (fp64) + the loss
                                                                           rebuilt from the fields by independent means
  configs[4]  ADMM, all four discretisations, at the per-GPU slab (32, 16, 1024, 1024) of the 256-plane volume:
                * every ADMM kernel on that slab with REAL neighbour halos == the unsharded call, and crops of it == oracle
                * the end-to-end iteration (2 outer x 3 CG) with the same kernel instantiations (M = 16, Nx = 1024,
                  32-plane slabs, production thresholds) on a volume the oracle can hold, two ranks with real halos
  README loop 1 (sub-gradient descent, README.md:118-124): all 300 iterations against the reference's golden loss.

Why crops are exact: one CP iteration reaches 2 voxels (D, then D^T); after n iterations a voxel depends on x0 within
radius 2n, so a crop that keeps the full z / t extent reproduces the full-volume result at every site at least 2n away
from the crop's artificial (y, x) borders.  ADMM's CG step sizes are GLOBAL dot products, so no crop can reproduce an
ADMM run: its end-to-end check uses a volume the oracle can run in full (see above).
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, PKG, ROOT, SCHEMES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pytv():
    import pytv as _p
    return _p


@pytest.fixture
def production(tvopt):
    """The library's own dispatch (other test modules lower the marching threshold for their small shapes)."""
    tvopt("TV_MARCH_MIN_PLANE_KB", 4096)
    tvopt("TV_FUSED_MIN_KVOXELS", 16384)
    tvopt("TV_ZCHUNK", 0)


def test_default_cp_path_follows_the_voxel_count(pytv, production):
    """solvers.ChambollePock(fused=None): the one-sweep path from 16 Mvoxel per slab on, whatever the plane size (round 1's
    plane-size rule kept BASELINE config 1 on the slower kernel pair); the kernel pair below (tools/archive/fused_vs_pair.py)."""
    import torch
    for shape, want in (((16, 4, 256, 256), False), ((32, 8, 256, 256), True), ((64, 1, 512, 512), True), ((8, 1, 1024, 1024), False)):
        cp = pytv.solvers.ChambollePock(torch.zeros(shape, device="cuda"), 25.0)
        assert cp.fused == want, (shape, cp.fused)
        del cp
    torch.cuda.empty_cache()


def _noisy_dev(shape, seed):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty(shape, device="cuda")
    for k in range(shape[0]):
        x[k] = 100.0 * torch.rand(shape[1:], device="cuda", generator=gen)
    # a piecewise-constant part so that the projection is active in some places and not in others
    x[shape[0] // 3:, :, shape[2] // 4: shape[2] // 2, shape[3] // 3:] += 120.0
    return x


# ------------------------------------------------------------------------------------------------
# configs[1]: 3-D 256 x 512 x 512, hybrid CP, whole volume against the C / OpenMP oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [None, False])
def test_config1_cp_full_volume_against_the_oracle(pytv, production, fused):
    import torch
    from oracle import tv_oracle_c as occ
    shape, n_it = (256, 1, 512, 512), 10
    x0 = _noisy_dev(shape, 11)
    cp = pytv.solvers.ChambollePock(x0, 25.0, scheme="hybrid", reg_z_over_reg=1.0, fused=fused)
    assert cp.fused == (fused is None)          # default at 67 Mvoxel: the one-sweep kernel (whatever the plane size); forced: the kernel pair
    loss = cp.run(n_it)
    wx, wloss = occ.chambolle_pock(x0.double().cpu().numpy(), n_it, 25.0, scheme="hybrid", reg_z_over_reg=1.0)
    np.testing.assert_allclose(loss, wloss, rtol=1e-5)
    np.testing.assert_allclose(cp.result().cpu().numpy(), wx, rtol=1e-5, atol=2e-3)


# ------------------------------------------------------------------------------------------------
# configs[2]: 4-D 128 x 8 x 512 x 512, hybrid CP: one-sweep and two-kernel paths
# ------------------------------------------------------------------------------------------------
def _loss_terms(pytv, x, x0, scheme, kw):
    """1/2 |x - x0|^2 and |D x|_{2,1} by means independent of the solver kernels: torch fp64 reductions over the
    materialised gradient (tv_D is pinned to the reference's golden vectors on its own)."""
    import torch
    from pytv import _native as nv
    fid = 0.0
    for k in range(0, x.shape[0], 16):
        fid += 0.5 * torch.sum((x[k:k + 16].double() - x0[k:k + 16].double()) ** 2).item()
    geo = nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, **kw)
    d = torch.empty(geo.grad_shape, device="cuda")
    nv.check(nv.lib().tv_D(geo.ref, nv.ptr(x), None, None, nv.ptr(d), nv.current_stream(x.device)))
    tv = 0.0
    for k in range(0, x.shape[0], 16):
        tv += torch.sqrt((d[k:k + 16].double() ** 2).sum(dim=1)).sum().item()
    return fid, tv


@pytest.mark.parametrize("fused", [None, False])
def test_config2_cp_crops_against_the_oracle(pytv, production, fused):
    import torch
    from oracle import tv_oracle_c as occ
    shape, n_it, reg = (128, 8, 512, 512), 10, 25.0
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    x0 = _noisy_dev(shape, 12)
    cp = pytv.solvers.ChambollePock(x0, reg, scheme="hybrid", fused=fused, **kw)
    assert cp.fused == (fused is None)          # 8 MiB planes: the one-sweep kernel is the default
    hist = torch.zeros((n_it, cp.SLOTS), dtype=torch.float64, device="cuda")
    x_prev = None
    for it in range(n_it):
        if it == n_it - 1:
            x_prev = cp.result().clone()
        cp.step(hist[it])
    loss = cp.loss_from_slots(hist.cpu().numpy(), reg)
    x, q = cp.result(), cp.q
    # (a) the loss of the last iteration, rebuilt: fidelity of x_n, TV of x_{n-1} (README.md:157 as the solver counts it)
    fid, _ = _loss_terms(pytv, x, x0, "hybrid", kw)
    _, tv_prev = _loss_terms(pytv, x_prev, x0, "hybrid", kw)
    assert abs(loss[-1] - (fid + reg * tv_prev)) <= 1e-6 * loss[-1]
    assert np.all(np.diff(loss) < 0)
    # (b) crops with the full z and t extent against the oracle; margin 2 * n_it from the artificial borders
    mg, cs = 2 * n_it, 96
    for (ya, xa) in [(0, 0), (512 - cs, 512 - cs), (200, 0), (130, 260)]:
        sub = x0[:, :, ya:ya + cs, xa:xa + cs].double().cpu().numpy()
        wx, wloss, wp, wq = _oracle_cp_state(occ, sub, n_it, reg, kw)
        ylo = 0 if ya == 0 else mg
        yhi = cs if ya + cs == 512 else cs - mg
        xlo = 0 if xa == 0 else mg
        xhi = cs if xa + cs == 512 else cs - mg
        got_x = x[:, :, ya + ylo:ya + yhi, xa + xlo:xa + xhi].cpu().numpy()
        np.testing.assert_allclose(got_x, wx[:, :, ylo:yhi, xlo:xhi], rtol=1e-5, atol=2e-3, err_msg="x crop (%d, %d)" % (ya, xa))
        got_q = q[:, :, :, ya + ylo:ya + yhi, xa + xlo:xa + xhi].cpu().numpy()
        np.testing.assert_allclose(got_q, wq[:, :, :, ylo:yhi, xlo:xhi], rtol=1e-5, atol=2e-3, err_msg="q crop (%d, %d)" % (ya, xa))


def _oracle_cp_state(occ, x0, n_it, reg, kw):
    """C / OpenMP oracle run that also hands back p and q (the wrapper's public function returns x and the loss)."""
    import ctypes
    g, keep = occ._geom(x0.shape, "hybrid", kw["reg_z_over_reg"], kw["reg_time"], False, 0)
    from oracle import tv_oracle as orc
    tau = orc.cp_step_size("hybrid", x0.shape[0], x0.shape[1], kw["reg_z_over_reg"], kw["reg_time"])
    x, p = x0.copy(), np.zeros_like(x0)
    q = np.zeros((x0.shape[0], g.nd) + x0.shape[1:], dtype=x0.dtype)
    d, dt = np.empty_like(q), np.empty_like(x0)
    loss = np.zeros(n_it)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    occ.lib().tvc_cp_f64(ctypes.byref(g), vp(x), vp(x0), vp(p), vp(q), vp(d), vp(dt), ctypes.c_int(n_it), ctypes.c_double(reg),
                         ctypes.c_double(0.5), ctypes.c_double(1.0), ctypes.c_double(tau), vp(loss))
    return x, loss, p, q


# ------------------------------------------------------------------------------------------------
# configs[4]: ADMM at the per-GPU slab (32, 16, 1024, 1024) of the 256-plane volume, four discretisations
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
def test_config4_admm_kernels_on_the_per_gpu_slab(pytv, production, scheme):
    """Planes [32, 64) of a 96-plane stand-in for the 256-plane volume (the kernels see nz_global = 256, z0 = 112 and
    the neighbours' planes as halos): every kernel of the ADMM iteration on the slab equals the same kernel on the
    whole 96 planes, bit for bit, and crops of the results equal the oracle."""
    import torch
    from oracle import tv_oracle as orc
    from pytv import _native as nv
    lib = nv.lib()
    m, ny, nx, nzs = 16, 1024, 1024, 32
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(21)
    full = (3 * nzs, m, ny, nx)
    x = torch.empty(full, device="cuda")
    for k in range(full[0]):
        x[k] = 50.0 * torch.rand(full[1:], device="cuda", generator=gen)
    gF = nv.Geometry(full, scheme, x.dtype, x.device, **kw)                                     # the 96 planes, unsharded
    gS = nv.Geometry((nzs,) + full[1:], scheme, x.dtype, x.device, nz_global=256, z0=112, **kw)  # the slab as rank 3 of 8 sees it
    st = nv.current_stream(x.device)
    nd = gF.nd
    per = 2 if scheme == "hybrid" else 1
    ch_b, ch_f = 2 * per, 2 * per + (1 if scheme == "hybrid" else 0)
    a, b = nzs, 2 * nzs
    ws = gF.workspace()
    sc = torch.zeros(4, dtype=torch.float64, device="cuda")
    ya, xa, cs = 500, 470, 48           # crop for the oracle (interior of the frame) and one at the frame corner
    crops = [(ya, xa), (0, 0), (ny - cs, nx - cs)]

    def crop(t, y0, x0_, lo, hi):
        return t[lo:hi, ..., y0:y0 + cs, x0_:x0_ + cs].double().cpu().numpy()

    def inner(arr, y0, x0_, r):
        """the part of a crop result not touched by the crop's artificial borders (r = stencil radius)"""
        sy = slice(0 if y0 == 0 else r, cs if y0 + cs == ny else cs - r)
        sx = slice(0 if x0_ == 0 else r, cs if x0_ + cs == nx else cs - r)
        return arr[..., sy, sx]

    # ---- z / u update: v = D x + u, z = shrink(v), u = v - z ------------------------------------------------
    u0 = torch.empty(gF.grad_shape, device="cuda")
    for k in range(full[0]):
        u0[k] = 20.0 * (torch.rand(gF.grad_shape[1:], device="cuda", generator=gen) - 0.5)
    zF, uF = torch.empty_like(u0), u0.clone()
    nv.check(lib.tv_admm_zu(gF.ref, nv.ptr(x), None, None, nv.ptr(zF), nv.ptr(uF), 7.5, sc[0:1].data_ptr(), nv.ptr(ws), st))
    zS, uS = torch.empty_like(u0[a:b]), u0[a:b].clone()
    nv.check(lib.tv_admm_zu(gS.ref, nv.ptr(x[a:b]), nv.ptr(x[a - 1:a]), nv.ptr(x[b:b + 1]), nv.ptr(zS), nv.ptr(uS), 7.5,
                            sc[1:2].data_ptr(), nv.ptr(gS.workspace()), st))
    assert torch.equal(zS, zF[a:b]) and torch.equal(uS, uF[a:b])
    for (y0, x0_) in crops:
        xs = crop(x, y0, x0_, a - 2, b + 2)
        us = crop(u0, y0, x0_, a - 2, b + 2)
        v = orc.D(xs, scheme, **kw) + us
        wz = orc.group_soft_threshold(v, 7.5)
        np.testing.assert_allclose(inner(crop(zF, y0, x0_, a, b), y0, x0_, 1), inner(wz[2:-2], y0, x0_, 1), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(inner(crop(uF, y0, x0_, a, b), y0, x0_, 1), inner((v - wz)[2:-2], y0, x0_, 1), rtol=1e-5, atol=1e-4)
    # ---- right-hand side: b = x0 + rho D^T (z - u) -----------------------------------------------------------
    bF = torch.empty_like(x)
    nv.check(lib.tv_DT_axpy(gF.ref, nv.ptr(zF), nv.ptr(uF), None, None, nv.ptr(x), 0.3, nv.ptr(bF), st))
    hp = (zF[a - 1, ch_b] - uF[a - 1, ch_b]).contiguous()
    hn = (zF[b, ch_f] - uF[b, ch_f]).contiguous()
    bS = torch.empty_like(x[a:b])
    nv.check(lib.tv_DT_axpy(gS.ref, nv.ptr(zF[a:b]), nv.ptr(uF[a:b]), nv.ptr(hp), nv.ptr(hn), nv.ptr(x[a:b]), 0.3, nv.ptr(bS), st))
    assert torch.equal(bS, bF[a:b])
    for (y0, x0_) in crops:
        ws_ = crop(zF, y0, x0_, a - 2, b + 2) - crop(uF, y0, x0_, a - 2, b + 2)
        want = crop(x, y0, x0_, a - 2, b + 2) + 0.3 * orc.D_T(ws_, scheme, **kw)
        np.testing.assert_allclose(inner(crop(bF, y0, x0_, a, b), y0, x0_, 2), inner(want[2:-2], y0, x0_, 2), rtol=1e-5, atol=2e-3)
    del zF, uF, zS, uS, u0, bS
    torch.cuda.empty_cache()
    # ---- normal operator (I + rho D^T D) and the CG vector updates ------------------------------------------------
    oF = torch.empty_like(x)
    nv.check(lib.tv_normal_op(gF.ref, nv.ptr(x), None, None, 0.3, nv.ptr(oF), sc[0:1].data_ptr(), nv.ptr(ws), st))
    oS = torch.empty_like(x[a:b])
    nv.check(lib.tv_normal_op(gS.ref, nv.ptr(x[a:b]), nv.ptr(x[a - 2:a]), nv.ptr(x[b:b + 2]), 0.3, nv.ptr(oS), sc[1:2].data_ptr(),
                              nv.ptr(gS.workspace()), st))
    assert torch.equal(oS, oF[a:b])
    dot_want = torch.sum(x[a:b].double() * oS.double()).item()
    assert abs(sc[1].item() - dot_want) <= 1e-9 * abs(dot_want)
    for (y0, x0_) in crops:
        xs = crop(x, y0, x0_, a - 3, b + 3)
        want = xs + 0.3 * orc.D_T(orc.D(xs, scheme, **kw), scheme, **kw)
        np.testing.assert_allclose(inner(crop(oF, y0, x0_, a, b), y0, x0_, 2), inner(want[3:-3], y0, x0_, 2), rtol=1e-5, atol=2e-3)
    # CG step on the slab: alpha = rs / dAd; x += alpha d; r -= alpha Ad; rs_new = <r, r>; d = r + beta d
    xs_, rs_, ds_ = bF[a:b].clone(), x[a:b].clone(), oS.clone()
    Ad = oF[a:b]
    sc[0], sc[1] = 3.0, 4.5
    alpha = np.float32(3.0 / 4.5)
    want_x = xs_ + alpha * ds_
    want_r = rs_ - alpha * Ad
    nv.check(lib.tv_cg_step1(gS.ref, nv.ptr(xs_), nv.ptr(rs_), nv.ptr(ds_), nv.ptr(Ad), sc[0:1].data_ptr(), sc[1:2].data_ptr(),
                             sc[2:3].data_ptr(), nv.ptr(gS.workspace()), st))
    assert torch.allclose(xs_, want_x, rtol=1e-6, atol=1e-5) and torch.allclose(rs_, want_r, rtol=1e-6, atol=1e-5)
    rr = torch.sum(rs_.double() ** 2).item()
    assert abs(sc[2].item() - rr) <= 1e-9 * rr
    want_d = rs_ + np.float32(rr / 3.0) * ds_
    nv.check(lib.tv_cg_step2(gS.ref, nv.ptr(ds_), nv.ptr(rs_), sc[2:3].data_ptr(), sc[0:1].data_ptr(), st))
    assert torch.allclose(ds_, want_d, rtol=1e-5, atol=1e-2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _admm_worker(rank, world, port, shape, scheme, kw, n_outer, n_cg, ret):
    import torch
    import torch.distributed as dist
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pytv
        from pytv.slab import Slab
        torch.cuda.set_device(0)
        rng = np.random.default_rng(33)
        x0_full = (60.0 * rng.random(shape)).astype(np.float32)
        slab = Slab(shape[0])
        x0 = torch.as_tensor(slab.local(x0_full).copy()).cuda()
        ad = pytv.solvers.ADMM(x0, 7.0, 0.1, n_cg=n_cg, scheme=scheme, slab=slab, keep_z=True, x_solver="cg", **kw)      # z is compared below
        assert ad.fused                      # the one-sweep dual side (tv_admm_fused + tv_admm_fixup) with real halos
        loss = ad.run(n_outer)
        win = (slice(None), slice(None), slice(8, 40), slice(300, 620))
        ret[rank] = dict(loss=loss, z=(slab.z0, slab.nz), x=ad.result()[win].cpu().numpy(),
                         zz=ad.z[:, :, :, 8:40, 300:620].cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scheme", SCHEMES)
def test_config4_admm_end_to_end_two_slabs_against_the_oracle(scheme, production):
    """2 outer x 3 CG iterations on (64, 16, 64, 1024): two ranks of 32 planes each (the per-GPU slab length of
    configs[4]) with real halos between them, M = 16 and Nx = 1024 as in configs[4] (the same kernel instantiations and,
    at 4 MiB planes, the production dispatch); the frame height is cut to 64 rows so that the C / OpenMP oracle can hold
    the whole volume -- the CG step sizes are global dot products, a crop cannot reproduce them."""
    import torch.multiprocessing as mp
    from oracle import tv_oracle_c as occ
    shape, n_outer, n_cg = (64, 16, 64, 1024), 2, 3
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_admm_worker, args=(2, _free_port(), shape, scheme, kw, n_outer, n_cg, ret), nprocs=2, join=True)
    assert len(ret) == 2
    rng = np.random.default_rng(33)
    x0 = (60.0 * rng.random(shape)).astype(np.float32).astype(np.float64)
    wx, wloss, wz, wu = occ.admm(x0, n_outer, 7.0, 0.1, n_cg, scheme=scheme, return_state=True, single_reduction=True, **kw)
    for r in range(2):
        z0, nz = ret[r]["z"]
        assert nz == 32
        # bounds = ~10 x what an fp32 run deviates from the fp64 oracle by on this problem (profiles/r3_admm_tolerances.txt,
        # tools/archive/admm_tolerance_probe.py: loss 3e-8 relative, x 1.1e-5 absolute at |x| <= 58, z 2.2e-5 at |z| <= 20 -- two or
        # three units in the last place of the largest values; round 2 had 5e-5 / 5e-3 here without a measurement behind them)
        np.testing.assert_allclose(ret[r]["loss"], wloss, rtol=1e-6)
        np.testing.assert_allclose(ret[r]["x"], wx[z0:z0 + nz, :, 8:40, 300:620], rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(ret[r]["zz"], wz[z0:z0 + nz, :, :, 8:40, 300:620], rtol=2e-6, atol=2e-4)


# ------------------------------------------------------------------------------------------------
# README loop 1 (configs[0] on the GPU): all 300 iterations against the reference's golden loss
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("one_pass", [True, False])
def test_subgradient_descent_300_iterations_against_the_reference_golden(pytv, scheme, one_pass):
    """tests/golden/trajectories_2d.npz holds the loss of the README's sub-gradient loop produced by the real reference
    (tests/golden/make_golden.py).  The head is compared tightly elsewhere; here the WHOLE trajectory with the bound
    the CPU oracle test uses for the tail (sub-gradient descent amplifies rounding differences at kinks)."""
    import torch
    z = np.load(os.path.join(GOLDEN, "trajectories_2d.npz"))
    noisy = z["noisy"]
    _, nb_it, reg, step = z["params"]
    want = z["gd_loss_" + scheme]
    x0 = torch.as_tensor(noisy).cuda()
    if one_pass:
        x0 = x0.float()                                     # the one-pass kernel is fp32
        if scheme == "central" and min(noisy.shape[:2]) == 2:
            pytest.skip("central with a two-point axis runs the two-pass kernels")
    sg = pytv.solvers.SubgradientDescent(x0, float(reg), float(step), scheme=scheme, one_pass=one_pass)
    assert sg.one_pass == one_pass
    loss = sg.run(int(nb_it))
    assert len(loss) == len(want) == 300
    np.testing.assert_allclose(loss[:40], want[:40], rtol=2e-5 if one_pass else 1e-9)
    np.testing.assert_allclose(loss, want, rtol=1e-3)


@pytest.mark.parametrize("one_pass", [True, False])
def test_subgradient_descent_at_config0_size_against_the_reference_golden(pytv, one_pass):
    """BASELINE configs[0] at its size: 512 x 512, hybrid, 300 iterations -- the loss curve the REAL reference produced
    (tests/golden/trajectory_512.npz, make_golden.py gen_trajectory_512) against the device-resident solver, one-pass
    (fp32) and two-pass (fp64) kernels."""
    import torch
    from test_oracle_golden import load_trajectory_512
    z, noisy, nb_it, reg, step = load_trajectory_512()
    want = z["gd_loss_hybrid"]
    x0 = torch.as_tensor(noisy).cuda()
    if one_pass:
        x0 = x0.float()
    sg = pytv.solvers.SubgradientDescent(x0, reg, step, scheme="hybrid", one_pass=one_pass)
    assert sg.one_pass == one_pass
    loss = sg.run(nb_it)
    np.testing.assert_allclose(loss[:40], want[:40], rtol=2e-5 if one_pass else 1e-9)
    np.testing.assert_allclose(loss, want, rtol=1e-3)
    assert abs(float(sg.x.double().mean()) - float(z["gd_final_mean"])) < 1e-2

"""Full-size oracle parity of the kernels that became the DEFAULT paths in round 3 (round-3 verdict, "What's missing" #3: the
default path must be the tested path):

  * tv_admm_fused + tv_admm_fixup + tv_cheb_step on the configs[4] per-GPU slab (32, 16, 1024, 1024) with real halos -- u is
    16 GiB per slab, byte offsets beyond 2^32: slab == the 96-plane call bit for bit, three crops == oracle, four schemes
    (the reference names ADMM only, README.md:26,135)
  * the one-pass sub-gradient kernel k_subgrad_col (32-bit buffer offsets with hardware range checks) at the north-star size
    (256, 8, 1024, 1024): G and the norms against oracle crops at planes >= 250, TV against tv_l21(tv_D(x))
    (pytv/tv_CPU.py:91-126 and its three siblings)
  * the fp64 one-sweep Chambolle-Pock iteration and the weight-volume sweep once each with more than 2^32 bytes of q
"""
import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pytv():
    import pytv as _p
    return _p


@pytest.fixture
def production(tvopt):
    tvopt("TV_MARCH_MIN_PLANE_KB", 4096)
    tvopt("TV_FUSED_MIN_KVOXELS", 16384)
    tvopt("TV_ZCHUNK", 0)


def _rand_planes(shape, scale, gen, offset=0.0):
    import torch
    t = torch.empty(shape, device="cuda")
    for k in range(shape[0]):
        t[k] = scale * (torch.rand(shape[1:], device="cuda", generator=gen) - offset)
    return t


# ------------------------------------------------------------------------------------------------
# configs[4]: the one-sweep ADMM dual side and the Chebyshev step on the per-GPU slab
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
def test_config4_admm_one_sweep_kernels_on_the_per_gpu_slab(pytv, production, tvopt, scheme):
    """Planes [32, 64) of a 96-plane stand-in for the 256-plane volume, as rank 3 of 8 sees them (nz_global = 256, z0 = 112,
    the neighbours' planes as halos).  The z-chunk length is pinned to the slab's (8 planes) for the 96-plane call so that both
    calls leave the same terms to the fix-up and the comparison can be bit for bit."""
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    m, ny, nx, nzs = 16, 1024, 1024, 32
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(41)
    full = (3 * nzs, m, ny, nx)
    x = _rand_planes(full, 50.0, gen)
    x0 = _rand_planes(full, 50.0, gen)
    gF = nv.Geometry(full, scheme, x.dtype, x.device, **kw)
    gS = nv.Geometry((nzs,) + full[1:], scheme, x.dtype, x.device, nz_global=256, z0=112, **kw)
    assert lib.tv_cp_fused_supported(gS.ref) == 1
    zc = lib.tv_cp_zchunk(gS.ref)
    assert zc == 8
    tvopt("TV_ZCHUNK", zc)
    assert lib.tv_cp_zchunk(gF.ref) == zc
    st = nv.current_stream(x.device)
    per = 2 if scheme == "hybrid" else 1
    ch_b, ch_f = 2 * per, 2 * per + (1 if scheme == "hybrid" else 0)
    a, b = nzs, 2 * nzs
    thresh, rho = 7.5, 0.3
    sc = torch.zeros(8, dtype=torch.float64, device="cuda")
    cs = 48
    crops = [(500, 470), (0, 0), (ny - cs, nx - cs)]

    def crop(t, y0, x0_, lo, hi):
        return t[lo:hi, ..., y0:y0 + cs, x0_:x0_ + cs].double().cpu().numpy()

    def inner(arr, y0, x0_, r):
        sy = slice(0 if y0 == 0 else r, cs if y0 + cs == ny else cs - r)
        sx = slice(0 if x0_ == 0 else r, cs if x0_ + cs == nx else cs - r)
        return arr[..., sy, sx]

    u0 = _rand_planes(gF.grad_shape, 20.0, gen, 0.5)
    assert u0[a:b].numel() * 4 > 2 ** 32                      # the slab's u: byte offsets beyond 32 bits
    u_crops = [crop(u0, y0, x0_, a - 2, b + 2) for (y0, x0_) in crops]      # the oracle's input, taken before u is updated in place
    uS, uS2 = u0[a:b].clone(), u0[a:b].clone()
    # ---- the 96 planes, unsharded: sweep + fix-up, every sample of t' stored (u updated in place) ------------------------------
    uF, tF, rF = u0, torch.zeros_like(u0), torch.empty_like(x)
    wsF = gF.workspace()
    nv.check(lib.tv_admm_fused(gF.ref, nv.ptr(x), None, None, nv.ptr(uF), nv.ptr(tF), nv.ptr(x0), nv.ptr(rF), thresh, rho, 1, 0, -1,
                               sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(wsF), st))
    nv.check(lib.tv_admm_fixup(gF.ref, nv.ptr(tF), None, None, nv.ptr(rF), rho, 0, -1, sc[2:3].data_ptr(), nv.ptr(wsF), st))
    # ---- the slab with real halos ------------------------------------------------------------------------------------------------
    tS, rS = torch.zeros_like(uS), torch.empty_like(x[a:b])
    wsS = gS.workspace()
    nv.check(lib.tv_admm_fused(gS.ref, nv.ptr(x[a:b]), nv.ptr(x[a - 1:a]), nv.ptr(x[b:b + 1]), nv.ptr(uS), nv.ptr(tS), nv.ptr(x0[a:b]),
                               nv.ptr(rS), thresh, rho, 1, 0, -1, sc[3:4].data_ptr(), sc[4:5].data_ptr(), nv.ptr(wsS), st))
    nv.check(lib.tv_admm_fixup(gS.ref, nv.ptr(tS), nv.ptr(tF[a - 1, ch_b]), nv.ptr(tF[b, ch_f]), nv.ptr(rS), rho, 0, -1,
                               sc[5:6].data_ptr(), nv.ptr(wsS), st))
    assert torch.equal(uS, uF[a:b]) and torch.equal(tS, tF[a:b])
    assert torch.equal(rS, rF[a:b])
    # <r, r> of the slab: sweep part + fix-up part == the direct sum
    rr = torch.sum(rS.double() ** 2).item()
    assert abs((sc[4] + sc[5]).item() - rr) <= 1e-9 * rr
    # TV of the slab's planes: the same number from both calls' partial sums is not available (the 96-plane call sums all planes);
    # against torch: |D x|_{2,1} of the slab from the oracle crops is checked through u / t' below
    del tS
    torch.cuda.empty_cache()
    # the sparse form (what ADMM(keep_z=False) runs) leaves the same u and r
    tS2, rS2 = torch.zeros_like(uS2), torch.empty_like(x[a:b])
    nv.check(lib.tv_admm_fused(gS.ref, nv.ptr(x[a:b]), nv.ptr(x[a - 1:a]), nv.ptr(x[b:b + 1]), nv.ptr(uS2), nv.ptr(tS2), nv.ptr(x0[a:b]),
                               nv.ptr(rS2), thresh, rho, 0, 0, -1, sc[6:7].data_ptr(), sc[7:8].data_ptr(), nv.ptr(wsS), st))
    nv.check(lib.tv_admm_fixup(gS.ref, nv.ptr(tS2), nv.ptr(tF[a - 1, ch_b]), nv.ptr(tF[b, ch_f]), nv.ptr(rS2), rho, 0, -1,
                               sc[5:6].data_ptr(), nv.ptr(wsS), st))
    assert torch.equal(uS2, uS) and torch.equal(rS2, rS)
    assert sc[6].item() == sc[3].item()                         # the TV partial sums do not depend on what is stored
    del uS2, tS2, rS2, uS
    torch.cuda.empty_cache()
    # ---- crops against the oracle ------------------------------------------------------------------------------------------------
    for (y0, x0_), us in zip(crops, u_crops):
        xs = crop(x, y0, x0_, a - 2, b + 2)
        dx = orc.D(xs, scheme, **kw)
        v = dx + us
        wz = orc.group_soft_threshold(v, thresh)
        wu = v - wz
        wt = (wz - wu) - dx
        wr = (crop(x0, y0, x0_, a - 2, b + 2) - xs) + rho * orc.D_T(wt, scheme, **kw)
        np.testing.assert_allclose(inner(crop(uF, y0, x0_, a, b), y0, x0_, 1), inner(wu[2:-2], y0, x0_, 1), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(inner(crop(tF, y0, x0_, a, b), y0, x0_, 1), inner(wt[2:-2], y0, x0_, 1), rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(inner(crop(rF, y0, x0_, a, b), y0, x0_, 2), inner(wr[2:-2], y0, x0_, 2), rtol=1e-5, atol=2e-3)
    del uF, tF, u0
    torch.cuda.empty_cache()
    # ---- one step of the Chebyshev x-solve: out = add + v + alpha (b - A v) + beta (v - y), A = I + rho D^T D -----------------------
    v, bb, yy = rF, x, x0                                  # three images at hand
    alpha, beta = 0.37, 0.21
    oF = torch.empty_like(x)
    nv.check(lib.tv_cheb_step(gF.ref, nv.ptr(v), None, None, rho, nv.ptr(bb), nv.ptr(yy), 0.0, nv.ptr(x0), None, alpha, beta, nv.ptr(oF),
                              sc[0:2].data_ptr(), nv.ptr(wsF), st))
    oS = torch.empty_like(x[a:b])
    nv.check(lib.tv_cheb_step(gS.ref, nv.ptr(v[a:b]), nv.ptr(v[a - 2:a]), nv.ptr(v[b:b + 2]), rho, nv.ptr(bb[a:b]), nv.ptr(yy[a:b]), 0.0,
                              nv.ptr(x0[a:b]), None, alpha, beta, nv.ptr(oS), sc[2:4].data_ptr(), nv.ptr(wsS), st))
    assert torch.equal(oS, oF[a:b])
    vv = torch.sum(v[a:b].double() ** 2).item()
    assert abs(sc[3].item() - vv) <= 1e-9 * vv                       # dots[1] = |v|^2 when no ref is given
    for (y0, x0_) in crops:
        vs = crop(v, y0, x0_, a - 3, b + 3)
        Av = vs + rho * orc.D_T(orc.D(vs, scheme, **kw), scheme, **kw)
        want = crop(x0, y0, x0_, a - 3, b + 3) + vs + alpha * (crop(bb, y0, x0_, a - 3, b + 3) - Av) + beta * (vs - crop(yy, y0, x0_, a - 3, b + 3))
        np.testing.assert_allclose(inner(crop(oF, y0, x0_, a, b), y0, x0_, 2), inner(want[3:-3], y0, x0_, 2), rtol=1e-5, atol=5e-3)


# ------------------------------------------------------------------------------------------------
# one-pass sub-gradient at the north-star size
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
def test_one_pass_subgradient_at_the_north_star_size(pytv, production, scheme):
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    shape = (256, 8, 1024, 1024)
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(43)
    x = _rand_planes(shape, 100.0, gen)
    x[80:, :, 256:512, 300:] += 120.0
    x[:, :, 700:710, :] = 55.0                    # flat patches: zero gradients (the |Dx| == 0 rule) inside the crops below
    g = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
    assert lib.tv_subgrad_fused_supported(g.ref) == 1
    st = nv.current_stream(x.device)
    ws = g.workspace()
    G, norms = torch.empty_like(x), torch.empty_like(x)
    sc = torch.zeros(4, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(G), sc[0:1].data_ptr(), nv.ptr(ws), st))
    G2 = torch.empty_like(x)
    nv.check(lib.tv_subgrad_fused_norms(g.ref, nv.ptr(x), None, None, nv.ptr(G2), nv.ptr(norms), sc[1:2].data_ptr(), nv.ptr(ws), st))
    assert torch.equal(G, G2)
    # TV against an independent route: tv_l21 over the materialised gradient (64 GiB)
    d = torch.empty(g.grad_shape, device="cuda")
    nv.check(lib.tv_D(g.ref, nv.ptr(x), None, None, nv.ptr(d), st))
    nv.check(lib.tv_l21(g.ref, nv.ptr(d), g.nd, None, sc[2:3].data_ptr(), nv.ptr(ws), st))
    del d
    torch.cuda.empty_cache()
    assert abs(sc[0].item() - sc[2].item()) <= 1e-6 * sc[2].item() and abs(sc[1].item() - sc[2].item()) <= 1e-6 * sc[2].item()
    # crops that end at the LAST plane of the volume: planes >= 250 (offsets beyond 2^33 bytes), two z planes of margin in front
    cs, lo = 64, 246
    for (y0, x0_) in [(0, 0), (1024 - cs, 1024 - cs), (672, 480), (300, 960)]:
        xs = x[lo:, :, y0:y0 + cs, x0_:x0_ + cs].double().cpu().numpy()
        _, wG, wn = orc.tv(xs, scheme, return_grad_norms=True, **kw)
        sy = slice(0 if y0 == 0 else 2, cs if y0 + cs == 1024 else cs - 2)
        sx = slice(0 if x0_ == 0 else 2, cs if x0_ + cs == 1024 else cs - 2)
        got = G[lo + 4:, :, y0:y0 + cs, x0_:x0_ + cs].cpu().numpy()
        np.testing.assert_allclose(got[..., sy, sx], wG[4:][..., sy, sx], rtol=1e-5, atol=1e-5, err_msg="G crop (%d, %d)" % (y0, x0_))
        gn = norms[lo + 4:, :, y0:y0 + cs, x0_:x0_ + cs].cpu().numpy()
        wn4 = wn[4:]
        s1 = slice(0 if y0 == 0 else 1, cs if y0 + cs == 1024 else cs - 1), slice(0 if x0_ == 0 else 1, cs if x0_ + cs == 1024 else cs - 1)
        fin = np.isfinite(wn4[..., s1[0], s1[1]])
        assert np.array_equal(np.isfinite(gn[..., s1[0], s1[1]]), fin)
        np.testing.assert_allclose(gn[..., s1[0], s1[1]][fin], wn4[..., s1[0], s1[1]][fin], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# fp64 one-sweep CP and the weight-volume sweep with q beyond 2^32 bytes
# ------------------------------------------------------------------------------------------------
def _cp_crops_against_oracle(cp, x0, n_it, reg, scheme, kw, W=None):
    from oracle import tv_oracle_c as occ
    nz, m, ny, nx = x0.shape
    loss = cp.run(n_it)
    assert np.all(np.diff(loss) < 0)
    x = cp.result()
    mg, cs = 2 * n_it, 64
    for (ya, xa) in [(0, 0), (ny - cs, nx - cs), (ny // 2, nx // 3)]:
        sub = x0[:, :, ya:ya + cs, xa:xa + cs].double().cpu().numpy()
        kws = dict(kw)
        if W is not None:
            kws["mask_static"] = W[:, :, ya:ya + cs, xa:xa + cs]
            wx, _ = orc.chambolle_pock(sub, n_it, reg, scheme=scheme, tau=cp.tau, **kws)
        else:
            wx, _ = occ.chambolle_pock(sub, n_it, reg, scheme=scheme, tau=cp.tau, **kws)
        sy = slice(0 if ya == 0 else mg, cs if ya + cs == ny else cs - mg)
        sx = slice(0 if xa == 0 else mg, cs if xa + cs == nx else cs - mg)
        got = x[:, :, ya:ya + cs, xa:xa + cs].double().cpu().numpy()
        tol = dict(rtol=1e-5, atol=2e-3) if x0.dtype.itemsize == 4 else dict(rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(got[..., sy, sx], wx[..., sy, sx], err_msg="crop (%d, %d)" % (ya, xa), **tol)


@pytest.mark.parametrize("scheme", ["hybrid", "central"])
def test_fp64_one_sweep_cp_with_q_beyond_4_gib(pytv, production, scheme):
    import torch
    shape = (20 if scheme == "hybrid" else 40, 8, 512, 1024)      # Nd = 8 / 4 channels: more than 2^32 bytes of q either way
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(44)
    x0 = _rand_planes(shape, 100.0, gen).double()
    x0[8:, :, 128:256, 300:] += 120.0
    cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, fused=True, **kw)
    assert cp.q.numel() * 8 > 2 ** 32
    _cp_crops_against_oracle(cp, x0, 4, 25.0, scheme, kw)


def test_weight_volume_sweep_with_q_beyond_4_gib(pytv, production):
    import torch
    shape = (36, 8, 512, 1024)                                     # q: 36 x 8 x 8 x 2 MiB = 4.5 GiB
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(45)
    x0 = _rand_planes(shape, 100.0, gen)
    x0[8:, :, 128:256, 300:] += 120.0
    rng = np.random.default_rng(45)
    W = (rng.random(shape) * 2.0).astype(np.float32)
    cp = pytv.solvers.ChambollePock(x0, 25.0, scheme="hybrid", fused=True, mask_static=W, **kw)
    assert cp.q.numel() * 4 > 2 ** 32 and cp.geo.weight_vol is not None
    _cp_crops_against_oracle(cp, x0, 3, 25.0, "hybrid", kw, W=W.astype(np.float64))


# ------------------------------------------------------------------------------------------------
# round-4 verdict, "missing" #3: the DEFAULT one-sweep Chambolle-Pock solver DIRECTLY against the oracle at the shape the
# headline number is quoted on, (256, 8, 1024, 1024), and at BASELINE configs[3], (512, 8, 1024, 1024) -- not through a second HIP path
# ------------------------------------------------------------------------------------------------
def _default_cp_crops_at_full_size(pytv, shape, n_it, seed, crops, cs, tune):
    """n_it iterations of solvers.ChambollePock as it comes (one sweep + fix-up; placement tuner as `tune`), then x AND q on (y, x)
    crops that keep the FULL z and t extent against the C / OpenMP oracle run on the crop (the method of
    tests/test_gpu_configs.py::test_config2_cp_crops_against_the_oracle: n iterations reach 2 n voxels, so sites 2 n away from the
    crop's artificial borders are exact).  Every plane -- q offsets beyond 2^33 / 2^34 bytes --, every z-chunk seam and every frame
    is inside each crop; the crops are placed on the block-tile column seams (255 | 256, 767 | 768), the wave-tile row seams
    (rows = 0, 7 mod 8) and two corners of the frame."""
    import torch
    from oracle import tv_oracle_c as occ
    from test_gpu_configs import _oracle_cp_state
    nz, m, ny, nx = shape
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x0 = _rand_planes(shape, 100.0, gen)
    x0[nz // 3:, :, ny // 4: ny // 2, nx // 5:] += 120.0          # the projection is active in some places and not in others
    cp = pytv.solvers.ChambollePock(x0, 25.0, scheme="hybrid", tune_placement=tune, **kw)
    assert cp.fused and cp.q.numel() * 4 >= 2 ** 36                 # one sweep; q of 64 GiB and more
    if tune:
        assert cp.placement is not None and "error" not in cp.placement, cp.placement
    loss = cp.run(n_it)
    assert np.all(np.diff(loss) < 0)
    x, q = cp.result(), cp.q
    mg = 2 * n_it
    for (ya, xa) in crops:
        sub = x0[:, :, ya:ya + cs, xa:xa + cs].double().cpu().numpy()
        wx, _wl, _wp, wq = _oracle_cp_state(occ, sub, n_it, 25.0, kw)
        sy = slice(0 if ya == 0 else mg, cs if ya + cs == ny else cs - mg)
        sx = slice(0 if xa == 0 else mg, cs if xa + cs == nx else cs - mg)
        got_x = x[:, :, ya:ya + cs, xa:xa + cs].cpu().numpy()
        np.testing.assert_allclose(got_x[..., sy, sx], wx[..., sy, sx], rtol=1e-5, atol=2e-3, err_msg="x crop (%d, %d)" % (ya, xa))
        got_q = q[:, :, :, ya:ya + cs, xa:xa + cs].cpu().numpy()
        np.testing.assert_allclose(got_q[..., sy, sx], wq[..., sy, sx], rtol=1e-5, atol=2e-3, err_msg="q crop (%d, %d)" % (ya, xa))
        # the deep planes on their own (a failure there would be an addressing bug, not arithmetic)
        for z in (nz - 6, nz - 1):
            np.testing.assert_allclose(got_q[z][..., sy, sx], wq[z][..., sy, sx], rtol=1e-5, atol=2e-3, err_msg="q plane %d" % z)
    del cp, x, q, x0
    torch.cuda.empty_cache()


def test_default_cp_at_the_north_star_size_against_the_oracle(pytv, production):
    """(256, 8, 1024, 1024): the tuned default solver (placement tuner ON, as bench.py runs it), 3 iterations."""
    _default_cp_crops_at_full_size(pytv, (256, 8, 1024, 1024), 3, 51, [(0, 0), (1024 - 40, 1024 - 40), (500, 236), (250, 748)], 40, None)


def test_default_cp_at_config3_size_against_the_oracle(pytv, production):
    """BASELINE configs[3] on ONE GPU (512, 8, 1024, 1024): x0, x, x_alt, p 16 GiB each + 128 GiB of q = 192 GiB; tuner off (memory)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < (200 << 30):
        pytest.skip("needs 200 GiB of free HBM")
    _default_cp_crops_at_full_size(pytv, (512, 8, 1024, 1024), 3, 52, [(0, 1024 - 40), (504, 236), (1024 - 40, 500)], 40, False)

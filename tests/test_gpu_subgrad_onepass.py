"""GPU parity of the ONE-PASS TV value + sub-gradient (C-ABI ``tv_subgrad_fused``, csrc/tv_subgrad.h) against the
CPU oracle (pytv/tv_CPU.py:47-375 restated), the golden vectors captured from the reference, and the two-pass
``tv_subgrad``.  fp32 kernel: ``rtol = atol = 1e-5`` against the fp64 oracle on the same up-cast input (the
reference's own bar, pytv/tests.py:88-109, and BASELINE.json's north_star tolerance)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"      # small test volumes take the one-sweep Chambolle-Pock path too

ONE_PASS_SCHEMES = ["upwind", "downwind", "hybrid", "central"]
F32 = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def pytv():
    import pytv
    return pytv


def _supported(x, scheme, **kw):
    from pytv import _native as nv
    geo = nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, **kw)
    return bool(nv.lib().tv_subgrad_fused_supported(geo.ref))


def _one_pass(pytv, x, scheme, **kw):
    tv, G, none = pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=False, one_pass=True, **kw)
    assert none is None
    return float(tv), G


# tile geometry of the kernel: 14 useful rows x 56 useful columns per block, ring of one row / one 4-column vector
@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
@pytest.mark.parametrize("zchunk", ["0", "3"])
@pytest.mark.parametrize("shape,lz,mu,use_mask", [
    ((1, 1, 16, 64), 1.0, 0.0, False),        # 2-D, one tile row, two tile columns
    ((1, 1, 33, 132), 0.0, 0.0, False),       # ragged in both directions
    ((5, 1, 14, 56), 1.0, 0.0, False),        # exactly one tile
    ((6, 1, 15, 60), 2.5, 0.0, False),        # one row / one vector more than a tile
    ((7, 2, 29, 116), 0.3, 0.7, False),
    ((4, 3, 17, 72), 1.0, 1.5, True),
    ((9, 4, 30, 64), 1.0, 2.0 ** -5, False),
    ((5, 8, 20, 128), 1.7, 0.6, True),
    ((4, 5, 17, 64), 1.0, 0.8, False),
    ((3, 6, 9, 72), 0.5, 1.2, True),
    ((5, 7, 15, 60), 1.0, 1.0, False),
    ((3, 9, 9, 64), 1.0, 1.0, False),          # M > 8: overlapping time windows of 8 frames (6 stored)
    ((4, 12, 17, 68), 0.7, 1.3, True),
    ((3, 16, 6, 64), 1.0, 0.5, False),
    ((2, 13, 5, 60), 0.0, 1.0, False),
    ((3, 25, 4, 64), 1.0, 1.0, False),
    ((3, 8, 9, 64), 0.0, 1.0, False),         # time axis only
    ((2, 2, 5, 68), 1.0, 1.0, False),
    ((3, 2, 6, 8), 1.0, 1.0, False),          # frame narrower than a tile
    ((2, 1, 3, 4), 1.0, 0.0, False),
    # Nx not a multiple of 4 (late round 3: the column-strip kernel holds one column per lane, so any Nx takes the one-pass path)
    ((3, 3, 17, 67), 1.3, 0.6, True),
    ((4, 1, 30, 125), 1.0, 0.0, False),
    ((2, 9, 5, 61), 1.0, 1.0, False),
    ((5, 2, 9, 7), 1.0, 0.5, False),
    ((1, 1, 21, 1001), 1.0, 0.0, False),
])
def test_one_pass_matches_oracle_and_two_pass(pytv, scheme, zchunk, shape, lz, mu, use_mask, tvopt):
    import torch
    tvopt("TV_ZCHUNK", zchunk)
    rng = np.random.default_rng(20 + shape[0] + shape[3])
    img = (rng.standard_normal(shape) * 10).astype(np.float32)
    img[:, :, 2:4, 2:20] = 3.0                  # a flat patch: |Dx| == 0 there (the 0 -> +inf rule)
    mask = (rng.random(shape[2:]) < 0.4) if use_mask else False
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=2.3 if use_mask else 0)
    x = torch.as_tensor(img).cuda()
    if not _supported(x, scheme, **kw):
        assert scheme == "central" and (shape[0] == 2 or shape[1] == 2)      # two-point axes: two-pass path only
        pytest.skip("central with a two-point axis")
    tv1, G1 = _one_pass(pytv, x, scheme, **kw)
    tv_ref, G_ref = orc.tv(img.astype(np.float64), scheme, **kw)
    np.testing.assert_allclose(G1.cpu().numpy(), G_ref, **F32)
    assert abs(tv1 - float(tv_ref)) <= 1e-6 * abs(float(tv_ref))
    tv2, G2, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, one_pass=False, **kw)
    np.testing.assert_allclose(G1.cpu().numpy(), G2.cpu().numpy(), rtol=2e-6, atol=2e-5)
    assert abs(tv1 - float(tv2)) <= 1e-6 * abs(tv1)


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
def test_one_pass_on_reference_golden(pytv, scheme):
    """Inputs / outputs captured from the real reference (tests/golden/make_golden.py), fp32 cases the kernel supports."""
    import torch
    from pytv import _native as nv
    z = np.load(os.path.join(GOLDEN, "ops_%s.npz" % scheme))
    done = 0
    for name in z["case_names"]:
        name = str(name)
        img = z[name + "/x"]
        lz, mu, factor = z[name + "/params"]
        mask = z[name + "/mask"]
        mask = False if mask.ndim == 0 else mask
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=factor)
        x = torch.as_tensor(img.astype(np.float32)).cuda()
        geo = nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, **kw)
        if not nv.lib().tv_subgrad_fused_supported(geo.ref):
            continue
        tv1, G1 = _one_pass(pytv, x, scheme, **kw)
        if img.dtype == np.float32:      # the reference's own fp32 outputs on this very input
            wtv, wG = float(z[name + "/tv"]), z[name + "/G"]
        else:                            # fp64 case: the kernel sees the fp32 rounding of the input
            wtv, wG = orc.tv(img.astype(np.float32).astype(np.float64), scheme, **kw)
        np.testing.assert_allclose(G1.cpu().numpy(), wG, err_msg="%s %s" % (scheme, name), **F32)
        assert abs(tv1 - float(wtv)) <= 1e-5 * abs(float(wtv)), (scheme, name)
        done += 1
    assert done >= 2      # every golden case the kernel supports (any Nx since late round 3)


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
@pytest.mark.parametrize("cuts", [(0, 3, 7), (0, 2, 4, 7), (0, 1, 2, 3, 4, 5, 6, 7)])
@pytest.mark.parametrize("zchunk", ["16", "2"])
def test_one_pass_slab_calls_equal_unsharded(pytv, scheme, cuts, zchunk, tvopt):
    import torch
    from pytv import _native as nv
    tvopt("TV_ZCHUNK", zchunk)
    lib = nv.lib()
    shape = (7, 3, 17, 132)
    rng = np.random.default_rng(5)
    kw = dict(reg_z_over_reg=1.7, reg_time=0.6)
    x = torch.as_tensor(rng.standard_normal(shape).astype(np.float32)).cuda()
    x0 = torch.as_tensor(rng.standard_normal(shape).astype(np.float32)).cuda()
    tv_full, G_full = _one_pass(pytv, x, scheme, **kw)
    # the fused descent step on the unsharded volume: x - step ((x - x0) + lam G)
    gfull = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
    xo_full, tvf, fidf = torch.empty_like(x), gfull.scalar(), gfull.scalar()
    nv.check(lib.tv_subgrad_step_fused(gfull.ref, nv.ptr(x), None, None, nv.ptr(x0), nv.ptr(xo_full), 0.05, 2.0, nv.ptr(tvf),
                                       nv.ptr(fidf), nv.ptr(gfull.workspace()), nv.current_stream(x.device)))
    want = x - 0.05 * ((x - x0) + 2.0 * G_full)
    assert torch.allclose(xo_full, want, rtol=1e-6, atol=1e-6)
    assert abs(float(tvf) - tv_full) <= 1e-9 * abs(tv_full)       # another instantiation: the compiler contracts differently
    assert abs(float(fidf) - 0.5 * float(((xo_full.double() - x0.double()) ** 2).sum())) <= 1e-6 * float(fidf)
    fid_sum = 0.0
    nzg, st, tv_sum = shape[0], nv.current_stream(x.device), 0.0
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = nv.Geometry((b - a,) + shape[1:], scheme, x.dtype, x.device, nz_global=nzg, z0=a, **kw)

        def two(lo):     # two-plane halo, NaN where the global plane does not exist (must never be read)
            buf = torch.full((2,) + shape[1:], float("nan"), dtype=x.dtype, device=x.device)
            for k in range(2):
                if 0 <= lo + k < nzg:
                    buf[k] = x[lo + k]
            return buf
        xp2 = two(a - 2) if a > 0 else None
        xn2 = two(b) if b < nzg else None
        xs = x[a:b].contiguous()
        G = torch.empty_like(xs)
        tvs = g.scalar()
        nv.check(lib.tv_subgrad_fused(g.ref, nv.ptr(xs), nv.ptr(xp2), nv.ptr(xn2), nv.ptr(G), nv.ptr(tvs), nv.ptr(g.workspace()), st))
        assert torch.equal(G, G_full[a:b]), (scheme, a, b)
        tv_sum += float(tvs)
        xo, fids = torch.empty_like(xs), g.scalar()
        nv.check(lib.tv_subgrad_step_fused(g.ref, nv.ptr(xs), nv.ptr(xp2), nv.ptr(xn2), nv.ptr(x0[a:b].contiguous()), nv.ptr(xo),
                                           0.05, 2.0, nv.ptr(tvs), nv.ptr(fids), nv.ptr(g.workspace()), st))
        assert torch.equal(xo, xo_full[a:b]), (scheme, a, b)
        fid_sum += float(fids)
    assert abs(tv_sum - tv_full) <= 1e-12 * abs(tv_full)
    assert abs(fid_sum - float(fidf)) <= 1e-12 * float(fidf)


def test_one_pass_rejects_what_it_does_not_support(pytv):
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    for shape, scheme, dt, kw in (((2, 1, 8, 64), "central", torch.float32, {}),                       # two-point z axis
                                  ((3, 2, 8, 64), "central", torch.float32, dict(reg_time=1.0)),       # two-point time axis
                                  ):
        g = nv.Geometry(shape, scheme, dt, torch.device("cuda", 0), **kw)
        assert lib.tv_subgrad_fused_supported(g.ref) == 0, (shape, scheme)
        x = torch.zeros(shape, dtype=dt, device="cuda")
        G = torch.empty_like(x)
        rc = lib.tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(g.scalar()), nv.ptr(g.workspace()),
                                  nv.current_stream(x.device))
        assert rc < 0
    # round 3: fp64 has a one-pass kernel of its own, and (late round 3) Nx may be anything, in both dtypes
    for shape, dt in (((2, 1, 8, 64), torch.float64), ((2, 1, 8, 65), torch.float64), ((2, 1, 8, 66), torch.float32), ((2, 1, 8, 7), torch.float32),
                      ((2, 16, 8, 66), torch.float32)):
        g = nv.Geometry(shape, "hybrid", dt, torch.device("cuda", 0), reg_time=1.0)
        assert lib.tv_subgrad_fused_supported(g.ref) == 1, shape
    # central with a two-point axis that is switched off is fine
    g = nv.Geometry((3, 2, 8, 64), "central", torch.float32, torch.device("cuda", 0), reg_time=0.0)
    assert lib.tv_subgrad_fused_supported(g.ref) == 1
    # a slab without its halos
    g = nv.Geometry((3, 1, 8, 64), "hybrid", torch.float32, torch.device("cuda", 0), nz_global=9, z0=3)
    x = torch.zeros((3, 1, 8, 64), device="cuda")
    rc = lib.tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(torch.empty_like(x)), nv.ptr(g.scalar()), nv.ptr(g.workspace()),
                              nv.current_stream(x.device))
    assert rc == -2


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
def test_one_pass_no_out_of_bounds_access(pytv, scheme):
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    shape = (4, 8, 19, 132)
    n = int(np.prod(shape))
    pad = 4100
    rng = np.random.default_rng(3)
    xb = torch.full((n + 2 * pad,), float("nan"), device="cuda")
    ob = torch.full((n + 2 * pad,), float("nan"), device="cuda")
    x = xb[pad:pad + n].view(shape)
    x.copy_(torch.as_tensor(rng.standard_normal(shape).astype(np.float32)).cuda())
    o = ob[pad:pad + n].view(shape)
    g = nv.Geometry(shape, scheme, torch.float32, torch.device("cuda", 0), reg_z_over_reg=1.2, reg_time=0.8)
    sc = g.scalar()
    nv.check(lib.tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(o), nv.ptr(sc), nv.ptr(g.workspace()), nv.current_stream(x.device)))
    torch.cuda.synchronize()
    assert bool(torch.isnan(ob[:pad]).all() and torch.isnan(ob[-pad:]).all())
    assert bool(torch.isfinite(o).all()) and np.isfinite(float(sc))


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
def test_subgradient_descent_one_pass_equals_two_pass(pytv, scheme):
    import torch
    shape = (6, 3, 20, 64)
    rng = np.random.default_rng(9)
    x0 = torch.as_tensor((rng.random(shape) * 50).astype(np.float32)).cuda()
    kw = dict(scheme=scheme, reg_z_over_reg=1.0, reg_time=0.5)
    a = pytv.solvers.SubgradientDescent(x0, 2.0, 0.05, one_pass=True, **kw)
    b = pytv.solvers.SubgradientDescent(x0, 2.0, 0.05, one_pass=False, **kw)
    la, lb = a.run(15), b.run(15)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    np.testing.assert_allclose(a.result().cpu().numpy(), b.result().cpu().numpy(), rtol=1e-4, atol=1e-3)
    ref_x, ref_loss = orc.subgradient_descent(x0.cpu().numpy().astype(np.float64), 15, 2.0, 0.05, **kw)
    np.testing.assert_allclose(la, ref_loss, rtol=1e-4)


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
def test_one_pass_error_on_the_readme_input_is_at_the_fp32_floor(pytv, scheme):
    """on rand(20,4,100,100) the reference's own fp32 run is within 8.7e-6 of its fp64 run in G.  The one-pass
    kernel uses v_rsq_f32 (1 ulp) for 1/|Dx|; its distance to the fp64 oracle must stay at that floor: measured 0.5-1.3e-6
    (IEEE two-pass kernel: 0.6-1.1e-6), bar 3e-6 here, 1e-5 in the north star."""
    import torch
    np.random.seed(0)
    img = np.random.rand(20, 4, 100, 100).astype(np.float32)
    kw = dict(reg_z_over_reg=1.0, reg_time=2.0 ** -5)
    tv_ref, G_ref = orc.tv(img.astype(np.float64), scheme, **kw)
    tv1, G1 = _one_pass(pytv, torch.as_tensor(img).cuda(), scheme, **kw)
    assert np.abs(G1.cpu().numpy() - G_ref).max() < 3e-6
    assert abs(tv1 - float(tv_ref)) < 1e-7 * float(tv_ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_subgradient_descent_graph_replay_equals_eager(pytv, dtype):
    """Small problems replay blocks of 10 iterations from a hipGraph: same trajectory as the eager loop."""
    import torch
    rng = np.random.default_rng(4)
    x0 = torch.as_tensor((rng.random((1, 1, 64, 64)) * 100).astype(dtype)).cuda()
    a = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3)
    b = pytv.solvers.SubgradientDescent(x0, 25.0, 5e-3)
    la, lb = a.run(47, graph=True), b.run(47, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result())


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
@pytest.mark.parametrize("shape,lz,mu", [((5, 3, 20, 64), 1.0, 0.5), ((3, 8, 9, 132), 1.3, 1.0), ((4, 12, 7, 64), 1.0, 0.7),
                                         ((1, 1, 33, 68), 1.0, 0.0)])
def test_one_pass_with_norms_matches_the_reference_convention(pytv, scheme, shape, lz, mu):
    """tv_subgrad_fused_norms: G, TV and grad_norms (|Dx| with zeros replaced by +inf, pytv/tv_GPU.py:88,135-139) from
    ONE pass; the drop-in tv_<scheme>(..., return_grad_norms=True) takes it."""
    import torch
    rng = np.random.default_rng(11 + shape[0])
    img = (rng.standard_normal(shape) * 10).astype(np.float32)
    img[:, :, 2:5, 2:30] = 3.0                  # a flat patch: |Dx| == 0 there -> +inf
    kw = dict(reg_z_over_reg=lz, reg_time=mu)
    if scheme == "central" and (shape[0] == 2 or shape[1] == 2):
        pytest.skip("central with a two-point axis")
    tv_ref, G_ref, n_ref = orc.tv(img.astype(np.float64), scheme, return_grad_norms=True, **kw)
    tv1, G1, n1 = pytv.tv_GPU.tv_subgradient_device(torch.as_tensor(img).cuda(), scheme, want_norms=True, one_pass=True, **kw)
    n1 = n1.cpu().numpy()
    assert np.array_equal(np.isinf(n1), np.isinf(n_ref)) and np.isinf(n1).sum() > 0
    fin = np.isfinite(n_ref)
    np.testing.assert_allclose(n1[fin], n_ref[fin], rtol=2e-6)
    np.testing.assert_allclose(G1.cpu().numpy(), G_ref, **F32)
    assert abs(float(tv1) - float(tv_ref)) <= 1e-6 * float(tv_ref)
    out = getattr(pytv.tv_GPU, "tv_" + scheme)(img.copy(), return_grad_norms=True, **kw)
    assert np.array_equal(out[2], n1) and np.array_equal(out[1], G1.cpu().numpy())


@pytest.mark.parametrize("scheme", ONE_PASS_SCHEMES)
def test_zero_gradient_rule_is_the_same_on_every_path(pytv, scheme):
    """A gradient whose SQUARE is below the smallest normal fp32 number (|Dx| < 1.1e-19) counts as zero on the one-pass
    and on the two-pass kernels alike (the reference zeroes at exactly 0; fp32 squares carry nothing below that), so
    return_grad_norms=True / False give the same G and TV.  Steps of 1e-21 on a flat image: every |Dx| is denormal-squared."""
    import torch
    shape = (3, 2, 8, 64)
    img = np.full(shape, 2.0, np.float32)
    img += (np.arange(64, dtype=np.float32) * 1e-30)[None, None, None, :]     # swallowed by fp32: exactly flat
    x = torch.as_tensor(img).cuda()
    tiny = torch.zeros(shape, device="cuda")
    tiny[:, :, 3, 10:20] = 1e-21              # |Dx| ~ 1e-21: its square underflows
    tiny[:, :, 5, 30:] = 1e-3                 # a regular step edge for contrast
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    res = []
    for one_pass in (True, False):
        if one_pass and scheme == "central" and min(shape[:2]) == 2:
            continue
        tv, G, n = pytv.tv_GPU.tv_subgradient_device(tiny, scheme, want_norms=True, one_pass=one_pass, **kw)
        res.append((float(tv), G.cpu().numpy(), n.cpu().numpy()))
    for tv, G, n in res:
        assert np.isinf(n[:, :, 3, 12]).all()                 # counted as zero gradient
        assert np.isfinite(n[:, :, 5, 29:31]).any()
        assert np.abs(G[:, :, 3, 12:18]).max() == 0.0         # and contributes nothing
    if len(res) == 2:
        np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
        assert np.array_equal(np.isinf(res[0][2]), np.isinf(res[1][2]))

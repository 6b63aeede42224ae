"""GPU parity of the round-4 EXPERIMENTAL geometry of the one-pass TV + sub-gradient kernel (csrc/tv_subgrad3.h: a lane = 2 rows x
2 columns; opt-in with TV_SG_KERNEL=3 for fp32, even Nx, 8-byte aligned arrays; not the default: no faster) against the CPU oracle (pytv/tv_CPU.py:47-375 restated) and
against the round-3 kernel (TV_SG_KERNEL=2) on the same input: G, the TV value, the per-voxel norms, the fused descent step, on
shapes that hold interior tiles (FAST variant: Nx >= 251, Ny >= 30), column-border tiles and generic tiles, with masks / a
per-pixel time factor / a weight volume, on z-slabs with halo planes, with more than 8 frames and on pitched arrays."""
import os

import numpy as np
import pytest

from oracle import tv_oracle as orc

# round 5: k_subgrad_pair is no longer part of the product library (it lost its A/B, profiles/r4_sgpattern.txt): this module runs only
# against a VARIANT build that holds it -- TV_WITH_OLD_SG=1 TV_VARIANT=oldsg python3 pytv-4d_amd/build.py, then
# PYTV4D_LIB=pytv-4d_amd/pytv/libpytv4d_hip_oldsg.so TV_TEST_OLD_SG=1 python -m pytest tests/test_gpu_subgrad_pair.py -m gpu
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("TV_TEST_OLD_SG") != "1",
                                                  reason="k_subgrad_pair lives in csrc/variants: needs a TV_WITH_OLD_SG build (see the module header)")]
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"

SCHEMES = ["upwind", "downwind", "hybrid", "central"]
F32 = dict(rtol=1e-5, atol=1e-5)

SHAPES = [
    ((3, 2, 33, 260), 1.0, 0.7, False),       # 3 x 3 tiles: one interior tile, column-border and generic tiles around it
    ((5, 8, 47, 384), 1.7, 0.6, False),       # M = 8, four tile columns
    ((4, 3, 62, 252), 0.4, 1.2, True),        # mask: every tile generic
    ((6, 1, 30, 376), 2.5, 0.0, False),       # no time axis
    ((1, 5, 31, 254), 0.0, 1.0, False),       # no z axis
    ((3, 11, 31, 256), 1.0, 1.3, False),      # M > 8: time windows
    ((2, 2, 5, 6), 1.0, 1.0, False),          # narrower than a lane pair of tiles
    ((4, 4, 16, 124), 1.0, 0.5, False),       # exactly one tile wide
    ((3, 3, 15, 126), 1.0, 0.5, False),       # one lane more
]


def _geo(nv, x, scheme, **kw):
    return nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, **kw)


def _fused(nv, x, scheme, norms=False, **kw):
    import torch
    g = _geo(nv, x, scheme, **kw)
    G = torch.full_like(x, float("nan"))
    tv = g.scalar()
    if norms:
        N = torch.full_like(x, float("nan"))
        nv.check(nv.lib().tv_subgrad_fused_norms(g.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(N), nv.ptr(tv), nv.ptr(g.workspace()),
                                                 nv.current_stream(x.device)))
        return float(tv), G, N
    nv.check(nv.lib().tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(g.workspace()), nv.current_stream(x.device)))
    return float(tv), G, None


def _img(shape, seed):
    rng = np.random.default_rng(seed)
    img = (rng.standard_normal(shape) * 10).astype(np.float32)
    if shape[2] > 4 and shape[3] > 24:
        img[:, :, 2:4, 2:20] = 3.0              # a flat patch: |Dx| == 0 there (the 0 -> +inf rule, the zero-gradient branch)
    return img, rng


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("zchunk", ["0", "2"])
@pytest.mark.parametrize("shape,lz,mu,use_mask", SHAPES)
def test_pair_kernel_matches_oracle_and_round3_kernel(scheme, zchunk, shape, lz, mu, use_mask, tvopt):
    import torch
    from pytv import _native as nv
    tvopt("TV_ZCHUNK", zchunk)
    img, rng = _img(shape, 40 + shape[0] + shape[3])
    mask = (rng.random(shape[2:]) < 0.4) if use_mask else False
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=2.3 if use_mask else 0)
    x = torch.as_tensor(img).cuda()
    if not nv.lib().tv_subgrad_fused_supported(_geo(nv, x, scheme, **kw).ref):
        pytest.skip("central with a two-point axis")
    tvopt("TV_SG_KERNEL", 3)
    tv3, G3, N3 = _fused(nv, x, scheme, norms=True, **kw)
    tv3b, G3b, _ = _fused(nv, x, scheme, norms=False, **kw)
    tvopt("TV_SG_KERNEL", 2)
    tv2, G2, N2 = _fused(nv, x, scheme, norms=True, **kw)
    tv_ref, G_ref, N_ref = orc.tv(img.astype(np.float64), scheme, return_grad_norms=True, **kw)
    np.testing.assert_allclose(G3.cpu().numpy(), G_ref, **F32)
    assert torch.equal(G3, G3b) and tv3 == tv3b                          # with and without the norms: the same G, bit for bit
    assert abs(tv3 - float(tv_ref)) <= 1e-6 * abs(float(tv_ref))
    n3 = N3.cpu().numpy()
    assert np.array_equal(np.isinf(n3), np.isinf(N_ref))
    fin = np.isfinite(N_ref)
    np.testing.assert_allclose(n3[fin], N_ref[fin], **F32)
    # the two kernels do the same arithmetic on different tiles: a few ulp of the largest term apart
    np.testing.assert_allclose(G3.cpu().numpy(), G2.cpu().numpy(), rtol=2e-6, atol=2e-5)
    assert abs(tv3 - tv2) <= 1e-6 * abs(tv3)
    assert np.array_equal(np.isinf(n3), np.isinf(N2.cpu().numpy()))


def test_pair_kernel_is_the_one_that_runs(tvopt):
    """The round-4 kernel must be what an eligible call takes: an odd pointer offset (8-byte misaligned) falls back to the round-3
    kernel, and both must agree -- but only the aligned call may differ from TV_SG_KERNEL=2 in its last bits."""
    import torch
    from pytv import _native as nv
    shape = (3, 4, 33, 260)
    img, _ = _img(shape, 7)
    x = torch.as_tensor(img).cuda()
    kw = dict(reg_time=0.9)
    tvopt("TV_SG_KERNEL", 3)
    tv3, G3, _ = _fused(nv, x, "hybrid", **kw)
    tvopt("TV_SG_KERNEL", 2)
    tv2, G2, _ = _fused(nv, x, "hybrid", **kw)
    assert not torch.equal(G3, G2)              # different tiles, different summation order somewhere in 100k voxels
    np.testing.assert_allclose(G3.cpu().numpy(), G2.cpu().numpy(), rtol=2e-6, atol=2e-5)
    # a view that starts one float into a larger buffer: 4-byte aligned only -> the round-3 kernel even with TV_SG_KERNEL=3
    buf = torch.zeros(x.numel() + 1, device="cuda")
    xo = buf[1:].view(shape)
    xo.copy_(x)
    assert xo.data_ptr() % 8 == 4
    tvopt("TV_SG_KERNEL", 3)
    g = _geo(nv, xo, "hybrid", **kw)
    Go = torch.empty_like(x)
    tvo = g.scalar()
    nv.check(nv.lib().tv_subgrad_fused(g.ref, nv.ptr(xo), None, None, nv.ptr(Go), nv.ptr(tvo), nv.ptr(g.workspace()), nv.current_stream(x.device)))
    assert torch.equal(Go, G2) and float(tvo) == tv2


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", [(4, 3, 33, 260), (3, 8, 31, 252), (3, 10, 17, 128)])
@pytest.mark.parametrize("what", ["time_factor", "weight_volume"])
def test_pair_kernel_with_time_factor_and_weight_volume(scheme, shape, what, tvopt):
    import torch
    from pytv import _native as nv
    img, rng = _img(shape, 90 + shape[1])
    x = torch.as_tensor(img).cuda()
    # a FLOAT mask_static is a weight on the time regularisation: per pixel (Ny, Nx) or per voxel (Nz, M, Ny, Nx)
    W = (0.5 + rng.random(shape[2:] if what == "time_factor" else shape)) * 1.5
    kw = dict(reg_z_over_reg=1.1, reg_time=0.8, mask_static=W)
    res = {}
    for k in (3, 2):
        tvopt("TV_SG_KERNEL", k)
        g = nv.Geometry(shape, scheme, x.dtype, x.device, **kw)
        assert (g.weight_vol is not None) == (what == "weight_volume")
        G = torch.full_like(x, float("nan"))
        tv = g.scalar()
        nv.check(nv.lib().tv_subgrad_fused(g.ref, nv.ptr(x), None, None, nv.ptr(G), nv.ptr(tv), nv.ptr(g.workspace()), nv.current_stream(x.device)))
        res[k] = (float(tv), G.cpu().numpy())
    tv_ref, G_ref = orc.tv(img.astype(np.float64), scheme, **kw)
    np.testing.assert_allclose(res[3][1], G_ref, **F32)
    assert abs(res[3][0] - float(tv_ref)) <= 1e-6 * abs(float(tv_ref))
    np.testing.assert_allclose(res[3][1], res[2][1], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("cuts", [(0, 3, 7), (0, 1, 2, 3, 4, 5, 6, 7)])
@pytest.mark.parametrize("zchunk", ["16", "2"])
def test_pair_kernel_on_slabs_equals_the_whole_volume(scheme, cuts, zchunk, tvopt):
    """z-slabs with two halo planes per interior side (tv_geom z0 / nz_global): every slab's G == the rows of the whole-volume G,
    bit for bit; the TV values add up."""
    import torch
    from pytv import _native as nv
    tvopt("TV_ZCHUNK", zchunk)
    tvopt("TV_SG_KERNEL", 3)
    shape = (7, 3, 33, 260)
    img, _ = _img(shape, 5)
    x = torch.as_tensor(img).cuda()
    kw = dict(reg_z_over_reg=1.4, reg_time=0.6)
    tv_all, G_all, _ = _fused(nv, x, scheme, **kw)
    tv_sum = 0.0
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = nv.Geometry((b - a,) + shape[1:], scheme, x.dtype, x.device, nz_global=shape[0], z0=a, **kw)
        def halo(lo, hi):
            buf = torch.full((2,) + shape[1:], float("nan"), dtype=x.dtype, device=x.device)
            for k, z in enumerate(range(lo, hi)):
                if 0 <= z < shape[0]:
                    buf[k] = x[z]
            return buf
        xp, xn = halo(a - 2, a), halo(b, b + 2)
        xs = x[a:b].contiguous()
        G = torch.full_like(xs, float("nan"))
        tv = g.scalar()
        nv.check(nv.lib().tv_subgrad_fused(g.ref, nv.ptr(xs), nv.ptr(xp) if a > 0 else None, nv.ptr(xn) if b < shape[0] else None, nv.ptr(G),
                                           nv.ptr(tv), nv.ptr(g.workspace()), nv.current_stream(x.device)))
        assert torch.equal(G, G_all[a:b]), (a, b)
        tv_sum += float(tv)
    assert abs(tv_sum - tv_all) <= 1e-6 * abs(tv_all)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape", [(5, 3, 33, 260), (4, 8, 31, 252), (3, 9, 17, 64)])
def test_pair_kernel_descent_step(scheme, shape, tvopt):
    """tv_subgrad_step_fused (README.md:118-124: x <- x - step ((x - x0) + lambda G)): the new x, the TV value and the fidelity term
    against the oracle, and against the round-3 kernel."""
    import torch
    from pytv import _native as nv
    img, rng = _img(shape, 3 + shape[1])
    x0 = torch.as_tensor(img).cuda()
    x = x0 + torch.as_tensor((rng.standard_normal(shape) * 2).astype(np.float32)).cuda()
    if not nv.lib().tv_subgrad_fused_supported(_geo(nv, x, scheme, reg_time=0.9).ref):
        pytest.skip("central with a two-point axis")
    lam, step = 7.0, 0.02
    out = {}
    for k in (3, 2):
        tvopt("TV_SG_KERNEL", k)
        g = _geo(nv, x, scheme, reg_z_over_reg=1.2, reg_time=0.9)
        xo = torch.full_like(x, float("nan"))
        tv, fid = g.scalar(), g.scalar()
        nv.check(nv.lib().tv_subgrad_step_fused(g.ref, nv.ptr(x), None, None, nv.ptr(x0), nv.ptr(xo), step, lam, nv.ptr(tv), nv.ptr(fid),
                                                nv.ptr(g.workspace()), nv.current_stream(x.device)))
        out[k] = (xo.cpu().numpy(), float(tv), float(fid))
    x64, x064 = x.cpu().numpy().astype(np.float64), img.astype(np.float64)
    tv_ref, G_ref = orc.tv(x64, scheme, reg_z_over_reg=1.2, reg_time=0.9)
    want = x64 - step * ((x64 - x064) + lam * G_ref)
    np.testing.assert_allclose(out[3][0], want, rtol=1e-5, atol=2e-5)
    assert abs(out[3][1] - float(tv_ref)) <= 1e-6 * abs(float(tv_ref))
    assert abs(out[3][2] - 0.5 * float(np.sum((want - x064) ** 2))) <= 1e-5 * out[3][2]
    np.testing.assert_allclose(out[3][0], out[2][0], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_pair_kernel_on_pitched_arrays(scheme, tvopt):
    """Padded solver state (interface version 4): rows of 250 columns on a 256-element pitch, a frame pitch with room to spare.  The
    images == the dense call's, the pads stay zero."""
    import torch
    from pytv import _native as nv
    tvopt("TV_SG_KERNEL", 3)
    shape = (4, 3, 31, 250)
    img, _ = _img(shape, 11)
    x = torch.as_tensor(img).cuda()
    kw = dict(reg_z_over_reg=0.8, reg_time=1.1)
    tv_d, G_d, _ = _fused(nv, x, scheme, **kw)
    g = nv.Geometry(shape, scheme, x.dtype, x.device, row_pitch=256, frame_pitch=256 * 31 + 512, **kw)
    xp = g.new_image()
    xp.copy_(x)
    Gp = g.new_image()
    tv = g.scalar()
    nv.check(nv.lib().tv_subgrad_fused(g.ref, nv.ptr(xp), None, None, nv.ptr(Gp), nv.ptr(tv), nv.ptr(g.workspace()), nv.current_stream(x.device)))
    assert torch.equal(Gp, G_d) and float(tv) == tv_d
    n = 1 + sum((int(a) - 1) * int(b) for a, b in zip(Gp.shape, Gp.stride()))
    assert float(Gp.as_strided((n,), (1,)).double().abs().sum()) == float(Gp.double().abs().sum())      # nothing but the image columns was written

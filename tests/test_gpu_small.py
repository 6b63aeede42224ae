"""The persistent small-volume loops (round 6: csrc/tv_small.hip, C-ABI tv_small_cp / tv_small_subgrad_descent) against the CPU oracle, the
reference's golden trajectories and the ordinary per-iteration kernels.  The reference's own shapes: README.md:76-79 rand(20,4,100,100),
README.md:107-124 / 141-157 (300 iterations on a 2-D image), pytv/tests.py:48 (N = 100, Nz = 20).

Tolerances as everywhere: fp64 1e-10, fp32 1e-5 relative on the loss against the fp64 oracle on the up-cast input."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pytv():
    import pytv
    return pytv


def _noisy(shape, seed, dtype):
    truth = orc.phantom(shape, seed=seed, dtype=np.float64)
    rng = np.random.RandomState(seed)
    return (truth + 100.0 * rng.rand(*shape)).astype(dtype)


FORMS = ["registers", "streamed2", "streamed3", "streamed4", "generic"]


def _set_form(nv, form):
    nv.set_option("TV_SMALL_GENERIC", 1 if form == "generic" else None)
    nv.set_option("TV_SMALL_SITES", int(form[-1]) if form and form.startswith("streamed") else None)


CASES = [((1, 1, 16, 16), 1.0, 0.0, False), ((6, 1, 16, 16), 1.0, 0.0, False), ((5, 3, 12, 16), 1.0, 1.0, False), ((4, 4, 9, 10), 2.5, 0.5, True),
         ((3, 2, 7, 13), 1.0, 1.0, False), ((2, 5, 33, 20), 0.0, 1.0, False), ((7, 2, 5, 70), 1.5, 0.25, True)]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape,lz,mu,use_mask", CASES)
def test_persistent_cp_matches_oracle(pytv, scheme, shape, lz, mu, use_mask, form):
    """the forms of the kernel: register-resident (one site-vector per thread), streamed (TV_SMALL_SITES: 2 .. 4 site-vectors per thread,
    state re-read every phase) and generic (TV_SMALL_GENERIC: the per-site bodies of the kernel pair in a loop); 16-byte lanes where Nx allows, scalar lanes otherwise (Nx = 10, 13: ragged rows)"""
    import torch
    from pytv import _native as nv
    rng = np.random.default_rng(4)
    mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=4.0 if use_mask else 0)
    _set_form(nv, form)
    try:
        for dtype, rtol, atol in ((np.float64, 1e-10, 1e-9), (np.float32, 1e-5, 2e-3)):
            x0 = _noisy(shape, 5, dtype)
            wx, wloss = orc.chambolle_pock(x0.astype(np.float64), 37, 25.0, scheme=scheme, **kw)
            cp = pytv.solvers.ChambollePock(torch.as_tensor(x0).cuda(), 25.0, scheme=scheme, persistent=True, pitch=None, **kw)
            assert cp.small and not cp.fused
            loss = cp.run(37)
            np.testing.assert_allclose(loss, wloss, rtol=rtol, err_msg="%s %s" % (scheme, shape))
            np.testing.assert_allclose(cp.result().cpu().numpy(), wx, rtol=rtol, atol=atol)
    finally:
        _set_form(nv, None)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("shape,lz,mu,use_mask", CASES)
def test_persistent_descent_matches_oracle(pytv, scheme, shape, lz, mu, use_mask, form):
    import torch
    from pytv import _native as nv
    rng = np.random.default_rng(4)
    mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
    kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=4.0 if use_mask else 0)
    _set_form(nv, form)
    try:
        for dtype, rtol, atol in ((np.float64, 1e-9, 1e-8), (np.float32, 2e-5, 5e-3)):
            x0 = _noisy(shape, 5, dtype)
            wx, wloss = orc.subgradient_descent(x0.astype(np.float64), 21, 25.0, 5e-3, scheme=scheme, **kw)
            sg = pytv.solvers.SubgradientDescent(torch.as_tensor(x0).cuda(), 25.0, 5e-3, scheme=scheme, persistent=True, pitch=None, **kw)
            assert sg.small
            loss = sg.run(21)                    # an odd count: the iterate ends in the other buffer of the ping-pong
            np.testing.assert_allclose(loss, wloss, rtol=rtol, err_msg="%s %s" % (scheme, shape))
            np.testing.assert_allclose(sg.result().cpu().numpy(), wx, rtol=rtol, atol=atol)
    finally:
        _set_form(nv, None)


TINY = [(1, 1, 1, 1), (1, 1, 2, 2), (1, 1, 1, 7), (1, 1, 9, 1), (3, 1, 1, 1), (1, 5, 1, 3), (2, 2, 2, 2), (1, 1, 3, 130), (2, 3, 1, 4)]


@pytest.mark.parametrize("form", ["registers", "streamed2", "generic"])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_persistent_loops_on_degenerate_shapes(pytv, scheme, form):
    """one-voxel images, single rows / columns, two-point axes (central falls back to the forward stencil there): the shapes the reference's
    slicing handles implicitly (pytv/tv_GPU.py:47-139) and a persistent kernel has to get right at its borders"""
    import torch
    from pytv import _native as nv
    _set_form(nv, form)
    try:
        for shape in TINY:
            x0 = _noisy(shape, 11, np.float64)
            kw = dict(reg_z_over_reg=0.7, reg_time=1.3 if shape[1] > 1 else 0.0)
            wx, wloss = orc.chambolle_pock(x0, 9, 25.0, scheme=scheme, **kw)
            cp = pytv.solvers.ChambollePock(torch.as_tensor(x0).cuda(), 25.0, scheme=scheme, persistent=True, pitch=None, **kw)
            np.testing.assert_allclose(cp.run(9), wloss, rtol=1e-10, atol=1e-9, err_msg="CP %s %s" % (scheme, shape))
            np.testing.assert_allclose(cp.result().cpu().numpy(), wx, rtol=1e-10, atol=1e-9)
            wx, wloss = orc.subgradient_descent(x0, 6, 25.0, 5e-3, scheme=scheme, **kw)
            sg = pytv.solvers.SubgradientDescent(torch.as_tensor(x0).cuda(), 25.0, 5e-3, scheme=scheme, persistent=True, pitch=None, **kw)
            np.testing.assert_allclose(sg.run(6), wloss, rtol=1e-9, atol=1e-9, err_msg="SG %s %s" % (scheme, shape))
            np.testing.assert_allclose(sg.result().cpu().numpy(), wx, rtol=1e-9, atol=1e-8)
    finally:
        _set_form(nv, None)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_persistent_cp_300_iterations_against_the_reference_golden(pytv, scheme):
    """README.md:141-157 as the real reference ran it (tests/golden/trajectories_2d.npz); 300 iterations = two launches of SMALL_BLOCK = 128
    iterations and a tail of 44: the state carried from launch to launch is x, p, q in memory"""
    import torch
    z = np.load(os.path.join(GOLDEN, "trajectories_2d.npz"))
    noisy = z["noisy"]
    _, nb_it, reg, _ = z["params"]
    for dtype, rtol in ((np.float64, 1e-10), (np.float32, 1e-5)):
        cp = pytv.solvers.ChambollePock(torch.as_tensor(noisy.astype(dtype)).cuda(), reg, scheme=scheme, tau=1 / 9, persistent=True)
        assert cp.small
        loss = cp.run(int(nb_it))
        np.testing.assert_allclose(loss, z["cp_loss_" + scheme], rtol=rtol)
        atol = 1e-8 if dtype == np.float64 else 2e-3
        np.testing.assert_allclose(cp.result().cpu().numpy(), z["cp_final_" + scheme], rtol=rtol, atol=atol)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_persistent_descent_300_iterations_against_the_reference_golden(pytv, scheme):
    """README.md:107-124 as the real reference ran it; fp64 to 1e-9 on the head, the whole trajectory with the bound the oracle test uses"""
    import torch
    z = np.load(os.path.join(GOLDEN, "trajectories_2d.npz"))
    noisy = z["noisy"]
    _, nb_it, reg, step = z["params"]
    want = z["gd_loss_" + scheme]
    for dtype, head in ((np.float64, 1e-9), (np.float32, 2e-5)):
        sg = pytv.solvers.SubgradientDescent(torch.as_tensor(noisy.astype(dtype)).cuda(), float(reg), float(step), scheme=scheme, persistent=True)
        assert sg.small
        loss = sg.run(int(nb_it))
        np.testing.assert_allclose(loss[:40], want[:40], rtol=head)
        np.testing.assert_allclose(loss, want, rtol=1e-3)


def test_persistent_descent_at_config0_size_against_the_reference_golden(pytv):
    """BASELINE configs[0] at its size (512 x 512, hybrid, 300 iterations): the loss curve of the REAL reference (trajectory_512.npz)"""
    import torch
    from test_oracle_golden import load_trajectory_512
    z, noisy, nb_it, reg, step = load_trajectory_512()
    want = z["gd_loss_hybrid"]
    for dtype, head in ((np.float64, 1e-9), (np.float32, 2e-5)):
        sg = pytv.solvers.SubgradientDescent(torch.as_tensor(noisy.astype(dtype)).cuda(), reg, step, scheme="hybrid")
        assert sg.small                          # the automatic rule picks the persistent loop at this size
        loss = sg.run(nb_it)
        np.testing.assert_allclose(loss[:40], want[:40], rtol=head)
        np.testing.assert_allclose(loss, want, rtol=1e-3)
        assert abs(float(sg.result().double().mean()) - float(z["gd_final_mean"])) < 1e-2


@pytest.mark.parametrize("scheme", ["hybrid", "upwind", "central"])
def test_readme_shape_persistent_equals_the_kernel_pair(pytv, scheme):
    """README.md:76-79: (20, 4, 100, 100).  The persistent Chambolle-Pock loop computes what tv_cp_dual + tv_cp_primal compute, site for
    site: after 50 iterations x and q agree with the kernel pair to a few units in the last place (only FMA contraction differs)."""
    import torch
    rng = np.random.default_rng(0)
    x0 = torch.as_tensor((100.0 * rng.random((20, 4, 100, 100))).astype(np.float32)).cuda()
    kw = dict(reg_z_over_reg=1.0, reg_time=1.0)
    a = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, persistent=True, **kw)
    b = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, fused=False, **kw)
    assert a.small and not b.small and not b.fused
    la, lb = a.run(50), b.run(50, graph=False)
    np.testing.assert_allclose(la, lb, rtol=1e-6)
    np.testing.assert_allclose(a.result().cpu().numpy(), b.result().cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(a.q.cpu().numpy(), b.q.cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(a.p.cpu().numpy(), b.p.cpu().numpy(), rtol=0, atol=2e-3)


def test_automatic_rule_and_explicit_requests(pytv):
    """(other GPU test modules lower TV_FUSED_MIN_KVOXELS to 0 for the whole process so that small volumes take the one-sweep kernel: the
    rule is tested with the product's threshold)"""
    import torch
    from pytv import _native as nv
    saved = nv.get_option("TV_FUSED_MIN_KVOXELS", -1)
    nv.set_option("TV_FUSED_MIN_KVOXELS", 16384)
    try:
        _automatic_rule(pytv, torch)
    finally:
        nv.set_option("TV_FUSED_MIN_KVOXELS", None if saved < 0 else saved)


def _automatic_rule(pytv, torch):
    x_small = torch.rand((4, 2, 32, 32), device="cuda")
    x_mid = torch.rand((20, 4, 256, 256), device="cuda")         # 5.2 Mvoxel: above SMALL_MAX_VOXELS (and above the library's TV_SMALL_MAX_KVOXELS)
    x_2m = torch.rand((9, 4, 256, 256), device="cuda")           # 2.4 Mvoxel: the generic form of the persistent kernels
    kw = dict(reg_time=1.0)
    assert pytv.solvers.ChambollePock(x_small, 1.0, **kw).small
    assert pytv.solvers.SubgradientDescent(x_small, 1.0, 1e-3, **kw).small
    assert not pytv.solvers.ChambollePock(x_mid, 1.0, **kw).small
    assert not pytv.solvers.SubgradientDescent(x_mid, 1.0, 1e-3, **kw).small
    assert not pytv.solvers.ChambollePock(x_small, 1.0, fused=False, **kw).small              # an explicit kernel family is kept
    assert not pytv.solvers.SubgradientDescent(x_small, 1.0, 1e-3, one_pass=True, **kw).small
    assert not pytv.solvers.ChambollePock(x_small, 1.0, persistent=False, **kw).small
    assert pytv.solvers.ChambollePock(x_2m, 1.0, **kw).small and pytv.solvers.SubgradientDescent(x_2m, 1.0, 1e-3, **kw).small
    with pytest.raises(ValueError):
        pytv.solvers.ChambollePock(x_mid, 1.0, persistent=True, **kw)                        # outside tv_small_supported
    # run_steps and step share the state: persistent blocks and single kernel-pair steps can be mixed
    cp = pytv.solvers.ChambollePock(x_small * 100, 25.0, **kw)
    ref = pytv.solvers.ChambollePock(x_small * 100, 25.0, persistent=False, fused=False, **kw)
    rows = torch.zeros((5, cp.SLOTS), dtype=torch.float64, device="cuda")
    cp.run_steps(rows[:3])
    cp.step(rows[3])
    cp.run_steps(rows[4:5])
    want = ref.run(5, graph=False)
    np.testing.assert_allclose(cp.loss_from_slots(rows.cpu().numpy(), 25.0), want, rtol=1e-6)


def test_c_abi_argument_checks(pytv):
    import ctypes
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    x = torch.rand((3, 2, 8, 8), device="cuda")
    geo = nv.Geometry(tuple(x.shape), "hybrid", x.dtype, x.device, 1.0, 1.0, False, 0)
    assert lib.tv_small_supported(geo.ref) == 1
    assert lib.tv_small_workspace_bytes(geo.ref, 0) == 0 and lib.tv_small_workspace_bytes(geo.ref, 8) > 0
    ws = torch.zeros(lib.tv_small_workspace_bytes(geo.ref, 8) // 8 + 1, dtype=torch.float64, device="cuda")
    h = torch.zeros(16, dtype=torch.float64, device="cuda")
    st = nv.current_stream(x.device)
    assert lib.tv_small_cp(geo.ref, None, nv.ptr(x), nv.ptr(x), nv.ptr(x), 0.5, 25.0, 0.1, 1.0, 8, h.data_ptr(), 2, 1, nv.ptr(ws), st) == -1
    assert lib.tv_small_cp(geo.ref, nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 0.5, 0.0, 0.1, 1.0, 8, h.data_ptr(), 2, 1, nv.ptr(ws), st) == -1
    assert b"lambda" in lib.tv_last_error()
    assert lib.tv_small_cp(geo.ref, nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 0.5, 25.0, 0.1, 1.0, 0, h.data_ptr(), 2, 1, nv.ptr(ws), st) == -1
    assert lib.tv_small_cp(geo.ref, nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 0.5, 25.0, 0.1, 1.0, 2, h.data_ptr(), 2, 2, nv.ptr(ws), st) == -1         # fid offset outside the row
    assert lib.tv_small_subgrad_descent(geo.ref, nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 1e-3, 25.0, 2, h.data_ptr(), 2, 1, nv.ptr(ws), st) == -1
    assert b"ping-pong" in lib.tv_last_error()
    # a slab of a larger volume is refused (no halos in a persistent launch)
    slab = nv.Geometry(tuple(x.shape), "hybrid", x.dtype, x.device, 1.0, 1.0, False, 0, nz_global=6, z0=3)
    assert lib.tv_small_supported(slab.ref) == 0
    assert lib.tv_small_cp(slab.ref, nv.ptr(x), nv.ptr(x), nv.ptr(x), nv.ptr(x), 0.5, 25.0, 0.1, 1.0, 2, h.data_ptr(), 2, 1, nv.ptr(ws), st) == -1


def test_a_launch_whose_blocks_are_not_resident_together_fails_loudly_instead_of_hanging():
    """The blocks of a persistent launch wait for each other.  With 7 blocks of 256 threads per CU (TV_SMALL_BLOCKS_PER_CU=7; the product bound is 4)
    hipLaunchCooperativeKernel accepted a 1764-block launch of the scalar-lane instantiation that the hardware did not keep resident together:
    the random walk hung on it.  The kernels now abandon such a launch after ~2 s of polling: NaN history, RuntimeError -- in a child process
    with a hard timeout, so that a regression shows as a failure, not as a hung suite.  (Where the hardware does keep them resident, the
    run must simply be correct.)"""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import os, sys
sys.path.insert(0, os.path.join(%r, "pytv-4d_amd")); sys.path.insert(0, %r)
import numpy as np, torch, pytv
rng = np.random.default_rng(1)
x0 = torch.as_tensor((rng.standard_normal((7, 9, 55, 102)) * 30 + 50).astype(np.float32)).cuda()
kw = dict(reg_z_over_reg=0.3, reg_time=1.7)
ref = pytv.solvers.ChambollePock(x0, 25.0, scheme="central", fused=False, pitch=None, **kw).run(2, graph=False)
try:
    got = pytv.solvers.ChambollePock(x0, 25.0, scheme="central", persistent=True, pitch=None, **kw).run(2)
    assert np.allclose(got, ref, rtol=2e-5), (got, ref)
    print("RESIDENT_AND_CORRECT")
except RuntimeError as e:
    assert "abandoned" in str(e)
    print("ABANDONED_LOUDLY")
torch.cuda.synchronize()
""" % (ROOT, ROOT)
    env = dict(os.environ, TV_SMALL_BLOCKS_PER_CU="7")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.stdout[-800:], p.stderr[-1500:])
    assert "RESIDENT_AND_CORRECT" in p.stdout or "ABANDONED_LOUDLY" in p.stdout
    # and with the product's bound the same volume runs
    env = {k: v for k, v in os.environ.items() if k != "TV_SMALL_BLOCKS_PER_CU"}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "RESIDENT_AND_CORRECT" in p.stdout, (p.stdout[-800:], p.stderr[-1500:])

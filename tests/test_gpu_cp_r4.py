"""Round-4 forms of the one-sweep Chambolle-Pock iteration (README.md:141-157 of the reference), against the forms they replace:

  * q ping-pong (tv_cp_sweep with q_in != q_out): the same arithmetic on the same values -- bit-identical x, q, loss;
  * lagged fidelity (TV_CP_FID_OF_INPUT; the fix-up reads no x0): the same iterates bit for bit, the loss history equal to the
    last digits (the fidelity is one fp64 sum over all sites instead of two partial sums), and equal to the oracle's.
"""
import os

import numpy as np
import pytest

from conftest import SCHEMES
from oracle import tv_oracle as orc

pytestmark = pytest.mark.gpu
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(6, 3, 32, 64), (9, 8, 24, 256), (4, 11, 16, 64), (5, 2, 17, 128), (1, 1, 40, 320)])
def test_pingpong_and_lagged_fidelity_equal_the_classic_loop(scheme, dtype, shape):
    import torch
    import pytv
    rng = np.random.default_rng(3)
    x0 = (orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(dtype)
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    n = 7
    dev = torch.as_tensor(x0).cuda()
    # classic: in-place q, every step returns its own fidelity (step() outside run_steps)
    a = pytv.solvers.ChambollePock(dev, 20.0, scheme=scheme, fused=True, q_pingpong=False, **kw)
    hist = torch.zeros((n, a.SLOTS), dtype=torch.float64, device="cuda")
    for k in range(n):
        a.step(hist[k])
    la = a.loss_from_slots(hist.cpu().numpy(), 20.0)
    # round 4: ping-pong + lagged fidelity (run -> run_steps)
    b = pytv.solvers.ChambollePock(dev, 20.0, scheme=scheme, fused=True, q_pingpong=True, **kw)
    assert b.q_alt is not None
    lb = b.run(n)
    assert torch.equal(a.result(), b.result()) and torch.equal(a.q, b.q) and torch.equal(a.p, b.p)
    np.testing.assert_allclose(lb, la, rtol=1e-12)
    # in-place q with the lagged fidelity, and a run split in two blocks
    c = pytv.solvers.ChambollePock(dev, 20.0, scheme=scheme, fused=True, q_pingpong=False, **kw)
    lc = np.concatenate([c.run(3), c.run(n - 3)])
    assert torch.equal(a.result(), c.result()) and torch.equal(a.q, c.q)
    np.testing.assert_allclose(lc, la, rtol=1e-12)
    _, wloss = orc.chambolle_pock(x0.astype(np.float64), n, 20.0, scheme=scheme, **kw)
    np.testing.assert_allclose(lb, wloss, rtol=1e-5 if dtype == np.float32 else 1e-11)


def test_fixup_without_x0_returns_zero_fidelity():
    import torch
    from pytv import _native as nv
    lib = nv.lib()
    shape = (4, 2, 16, 64)
    g = nv.Geometry(shape, "hybrid", torch.float32, "cuda", reg_time=1.0)
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.random(shape).astype(np.float32)).cuda()
    x0 = torch.as_tensor(rng.random(shape).astype(np.float32)).cuda()
    q, q2, p, xo = torch.zeros(g.grad_shape, device="cuda"), torch.zeros(g.grad_shape, device="cuda"), torch.zeros_like(x), torch.empty_like(x)
    sc = torch.zeros(4, dtype=torch.float64, device="cuda")
    ws, st = g.workspace(), nv.current_stream(x.device)
    nv.check(lib.tv_cp_sweep(g.ref, nv.ptr(x), None, None, nv.ptr(q), nv.ptr(q2), nv.ptr(x0), nv.ptr(p), nv.ptr(xo), 0.5, 5.0, 0.05, 1.0, 1, 0, -1,
                             sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(ws), st))
    want = 0.5 * torch.sum((x.double() - x0.double()) ** 2).item()
    assert abs(sc[1].item() - want) <= 1e-12 * want                       # fidelity of the INPUT over all sites
    sc[2] = 7.0
    nv.check(lib.tv_cp_fixup(g.ref, nv.ptr(q2), None, None, nv.ptr(xo), None, 0.05, 0, -1, sc[2:3].data_ptr(), nv.ptr(ws), st))
    assert sc[2].item() == 0.0
    assert lib.tv_cp_sweep(g.ref, nv.ptr(x), None, None, nv.ptr(q), nv.ptr(q2), nv.ptr(x0), nv.ptr(p), nv.ptr(xo), 0.5, 5.0, 0.05, 1.0, 8, 0, -1,
                           sc[0:1].data_ptr(), sc[1:2].data_ptr(), nv.ptr(ws), st) == -1


@pytest.mark.parametrize("scheme", ["hybrid", "upwind"])
@pytest.mark.parametrize("shape", [(6, 3, 32, 64), (5, 3, 33, 250)])       # the second: 1000-byte rows, padded state (pitch="auto")
def test_placement_tuner_changes_nothing_but_the_buffers(scheme, shape):
    """ChambollePock(tune_placement=True) times candidate allocations for q, the x ping-pong pair and p with the real sweep and keeps
    the fastest; the state is re-initialised afterwards -- the run must be bit-identical to an untuned one."""
    import torch
    import pytv
    rng = np.random.default_rng(5)
    x0 = torch.as_tensor((orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(np.float32)).cuda()
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3)
    a = pytv.solvers.ChambollePock(x0, 20.0, scheme=scheme, fused=True, tune_placement=False, **kw)
    b = pytv.solvers.ChambollePock(x0, 20.0, scheme=scheme, fused=True, tune_placement=True, **kw)
    assert a.placement is None and b.placement is not None
    info = b.placement
    assert info["candidates"] == 4 and len(info["sweep_ms"]) == 4 and info["seconds"] > 0 and len(info["p_round_trip_ms"]) == 3
    la, lb = a.run(7), b.run(7)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result()) and torch.equal(a.q, b.q) and torch.equal(a.p, b.p)


@pytest.mark.parametrize("scheme", ["hybrid", "central"])
@pytest.mark.parametrize("pitch", ["auto", "dense"])
def test_subgradient_descent_placement_tuner_changes_nothing_but_the_buffers(scheme, pitch):
    """SubgradientDescent(tune_placement=True) times the ordered pairs of four candidate image buffers (and x0 against a copy) with the
    real one-pass step kernel and keeps the fastest; the iterate is re-initialised afterwards -- bit-identical to an untuned run."""
    import torch
    import pytv
    rng = np.random.default_rng(8)
    shape = (5, 3, 33, 250)                          # 1000-byte rows: pitch="auto" pads them
    x0 = torch.as_tensor((orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(np.float32)).cuda()
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3, scheme=scheme, pitch=pitch)
    a = pytv.solvers.SubgradientDescent(x0, 20.0, 5e-3, tune_placement=False, **kw)
    b = pytv.solvers.SubgradientDescent(x0, 20.0, 5e-3, tune_placement=True, **kw)
    assert a.one_pass and a.placement is None and b.placement is not None and "error" not in b.placement
    info = b.placement
    assert info["candidates"] == 4 and len(info["step_ms"]) == 4 and info["seconds"] > 0 and len(info["x0_round_trip_ms"]) == 2
    la, lb = a.run(9, graph=False), b.run(9, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result())


@pytest.mark.parametrize("scheme", ["hybrid", "upwind"])
@pytest.mark.parametrize("x_solver", ["chebyshev", "cg"])
def test_admm_placement_tuner_changes_nothing_but_the_buffers(scheme, x_solver):
    """ADMM(tune_placement=True) times complete sets of state arrays with real outer iterations and keeps the fastest, back in the initial
    state -- bit-identical to an untuned run (and ``z`` still available)."""
    import torch
    import pytv
    rng = np.random.default_rng(9)
    shape = (6, 3, 32, 64)
    x0 = torch.as_tensor((orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(np.float32)).cuda()
    kw = dict(reg_z_over_reg=0.7, reg_time=1.3, scheme=scheme, n_cg=4, x_solver=x_solver)
    a = pytv.solvers.ADMM(x0, 6.0, 0.1, tune_placement=False, **kw)
    b = pytv.solvers.ADMM(x0, 6.0, 0.1, tune_placement=True, **kw)
    assert a.placement is None and b.placement is not None and "error" not in b.placement
    assert len(b.placement["outer_ms"]) == 4 and b.placement["seconds"] > 0
    la, lb = a.run(5, graph=False), b.run(5, graph=False)
    assert np.array_equal(la, lb)
    assert torch.equal(a.result(), b.result()) and torch.equal(a.u, b.u) and torch.equal(a.z, b.z)


@pytest.mark.parametrize("shape,pitch", [((6, 3, 32, 64), "auto"), ((5, 3, 33, 250), "auto")])
def test_arena_option_changes_nothing_but_the_buffers(shape, pitch):
    """ChambollePock(arena=True) (round 5, opt-in): x, x_alt, p, a private copy of x0 and q carved out of ONE allocation -- the run must be
    bit-identical to the default (separate allocations), the caller's x0 untouched, dense and padded state alike."""
    import torch
    import pytv
    rng = np.random.default_rng(11)
    x0 = torch.as_tensor((orc.phantom(shape, dtype=np.float64) + 100 * rng.random(shape)).astype(np.float32)).cuda()
    keep = x0.clone()
    kw = dict(scheme="hybrid", reg_z_over_reg=0.7, reg_time=1.3, fused=True, pitch=pitch)
    a = pytv.solvers.ChambollePock(x0, 20.0, **kw)
    b = pytv.solvers.ChambollePock(x0, 20.0, arena=True, **kw)
    assert b.arena and b._arena is not None and not a.arena
    base = b._arena.data_ptr()
    for t in (b.x, b.x_alt, b.p, b.x0, b.q):
        assert base <= t.data_ptr() < base + b._arena.numel() * 4
    b._tune_arena()                                   # the two-candidate measurement leaves the initial state behind
    la, lb = a.run(6), b.run(6)
    assert torch.equal(a.result(), b.result()) and torch.equal(a.q, b.q) and torch.equal(a.p, b.p)
    np.testing.assert_allclose(lb, la, rtol=1e-13)
    assert torch.equal(x0, keep)

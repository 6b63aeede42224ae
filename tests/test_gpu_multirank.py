"""N > 1 on REAL kernels with ONE GPU: 2-4 processes share cuda:0, talk over gloo (halo planes
staged through the host -- pytv/slab.py does that automatically for the gloo backend), and run
the z-slab solvers end to end: halo plan, interior/edge launches with overlap, sub-slab geometries,
scalar all-reduce.  The result must equal the unsharded oracle run.  RCCL itself cannot put two
ranks on one device, so the device-to-device transport is exercised only by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, SCHEMES

pytestmark = pytest.mark.gpu

# exercise the plane-marching kernels on the small test shapes too (production threshold: 4 MiB planes)
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"      # small test volumes take the one-sweep Chambolle-Pock path too


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, scheme, kw, overlap, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pytv
        from pytv.slab import Slab
        torch.cuda.set_device(0)
        rng = np.random.default_rng(91)
        x0_full = (60.0 * rng.random(shape)).astype(np.float32)
        slab = Slab(shape[0])
        x0 = torch.as_tensor(slab.local(x0_full).copy()).cuda()
        out = {}
        # two-kernel path (interior-first overlap when asked) ...
        cp = pytv.solvers.ChambollePock(x0, 7.0, scheme=scheme, slab=slab, overlap=overlap, fused=False, **kw)
        out["cp_loss"] = cp.run(8)
        out["cp_x"] = cp.result().cpu().numpy()
        out["cp_overlap"] = cp.overlap
        # ... and the library's default (the one-sweep kernel where the geometry supports it)
        # (the placement tuner forced on: on a slab it launches local sweeps only -- every rank of a weak-scaling run tunes, round 4)
        cp2 = pytv.solvers.ChambollePock(x0, 7.0, scheme=scheme, slab=slab, tune_placement=True, **kw)
        out["cp2_tuned"] = cp2.placement is not None and "error" not in cp2.placement
        out["cp2_loss"] = cp2.run(8)
        out["cp2_x"] = cp2.result().cpu().numpy()
        out["cp2_fused"] = cp2.fused
        out["cp2_overlap"] = bool(getattr(cp2, "overlap_fused", False))
        if min(n for _, n in slab.parts) >= 2:
            sg = pytv.solvers.SubgradientDescent(x0, 7.0, 2e-3, scheme=scheme, slab=slab, tune_placement=True, **kw)
            out["sg_loss"] = sg.run(5)
            out["sg_x"] = sg.result().cpu().numpy()
            ad = pytv.solvers.ADMM(x0, 7.0, 0.1, n_cg=3, scheme=scheme, slab=slab, x_solver="cg", **kw)
            out["ad_loss"] = ad.run(3)
            out["ad_x"] = ad.result().cpu().numpy()
            ac = pytv.solvers.ADMM(x0, 7.0, 0.1, n_cg=4, scheme=scheme, slab=slab, **kw)      # the default x-solve (Chebyshev): no all-reduce
            assert ac.cheb
            out["ac_loss"] = ac.run(3)
            out["ac_x"] = ac.result().cpu().numpy()
        # data-fidelity operator slot on a slab: a diagonal operator (local to the slab), TV part with halos
        a_full = 0.2 + 0.8 * np.random.default_rng(92).random(shape)
        at = torch.as_tensor(slab.local(a_full).astype(np.float32).copy()).cuda()
        bt = at * x0 + 0.5
        op = pytv.solvers.ChambollePockOperator(lambda v: at * v, lambda v: at * v, bt, x0, 7.0, scheme=scheme, slab=slab, **kw)
        out["op_loss"] = op.run(6)
        out["op_x"] = op.result().cpu().numpy()
        out["z"] = (slab.z0, slab.nz)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("world,shape,overlap,zchunk", [(2, (8, 3, 6, 132), True, "1"), (3, (9, 2, 8, 16), True, "0"),
                                                        (4, (8, 4, 5, 128), False, "0"), (2, (12, 2, 9, 68), True, "2"),
                                                        (2, (14, 2, 5, 16), True, "0"),
                                                        (2, (8, 12, 5, 64), True, "1")])      # M > 8: time windows
def test_sharded_solvers_equal_unsharded_oracle(scheme, world, shape, overlap, zchunk, tvopt):
    from oracle import tv_oracle as orc
    tvopt("TV_ZCHUNK", zchunk)     # inherited by the spawned ranks; "1"/"2": >= 3 chunks per rank -> overlap
    kw = dict(reg_z_over_reg=1.3, reg_time=0.7)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, scheme, kw, overlap, ret), nprocs=world, join=True)
    assert len(ret) == world
    rng = np.random.default_rng(91)
    x0 = (60.0 * rng.random(shape)).astype(np.float32).astype(np.float64)
    wx, wloss = orc.chambolle_pock(x0, 8, 7.0, scheme=scheme, **kw)
    for r in range(world):
        z0, nz = ret[r]["z"]
        np.testing.assert_allclose(ret[r]["cp_loss"], wloss, rtol=1e-5, err_msg="rank %d" % r)
        np.testing.assert_allclose(ret[r]["cp_x"], wx[z0:z0 + nz], rtol=1e-5, atol=2e-3, err_msg="rank %d" % r)
        np.testing.assert_allclose(ret[r]["cp2_loss"], wloss, rtol=1e-5, err_msg="rank %d (default path)" % r)
        np.testing.assert_allclose(ret[r]["cp2_x"], wx[z0:z0 + nz], rtol=1e-5, atol=2e-3, err_msg="rank %d (default path)" % r)
    if shape[-1] % 4 == 0 and shape[-1] >= 64:
        assert all(ret[r]["cp2_fused"] for r in range(world))
        assert all(ret[r]["cp2_tuned"] for r in range(world))
        if overlap and zchunk in ("1", "2"):
            assert all(ret[r]["cp2_overlap"] for r in range(world))     # interior-first one-sweep path exercised
    if overlap:
        assert any(ret[r]["cp_overlap"] for r in range(world))
    # operator-slot solver against the same iteration written with the oracle's D / D^T
    a = 0.2 + 0.8 * np.random.default_rng(92).random(shape)
    a = a.astype(np.float32).astype(np.float64)
    b = a * x0 + 0.5
    tau = orc.cp_step_size(scheme, shape[0], shape[1], kw["reg_z_over_reg"], kw["reg_time"])
    x, p, q = x0.copy(), np.zeros(shape), np.zeros_like(orc.D(x0, scheme, **kw))
    want = []
    for _ in range(6):
        p = (p + 1.0 * (a * x - b)) / 2.0
        Dx = orc.D(x, scheme, **kw)
        v = q + 0.5 * Dx
        q = v / np.maximum(1.0, np.sqrt(np.sum(v ** 2, axis=1, keepdims=True)) / 7.0)
        x = x - tau * (a * p) - tau * orc.D_T(q, scheme, **kw)
        want.append(0.5 * np.sum((a * x - b) ** 2) + 7.0 * orc.compute_L21_norm(Dx))
    for r in range(world):
        z0, nz = ret[r]["z"]
        np.testing.assert_allclose(ret[r]["op_loss"], want, rtol=2e-5, err_msg="operator slot, rank %d" % r)
        np.testing.assert_allclose(ret[r]["op_x"], x[z0:z0 + nz], rtol=1e-4, atol=2e-3)
    if "sg_loss" in ret[0]:
        sx, sloss = orc.subgradient_descent(x0, 5, 7.0, 2e-3, scheme=scheme, **kw)
        ax, aloss = orc.admm(x0, 3, 7.0, 0.1, 3, scheme=scheme, single_reduction=True, **kw)
        for r in range(world):
            z0, nz = ret[r]["z"]
            np.testing.assert_allclose(ret[r]["sg_loss"], sloss, rtol=1e-5)
            np.testing.assert_allclose(ret[r]["sg_x"], sx[z0:z0 + nz], rtol=1e-5, atol=2e-3)
            np.testing.assert_allclose(ret[r]["ad_loss"], aloss, rtol=1e-6)          # ~10 x measured: profiles/r3_admm_tolerances.txt
            np.testing.assert_allclose(ret[r]["ad_x"], ax[z0:z0 + nz], rtol=2e-6, atol=2e-4)
        cx, closs = orc.admm(x0, 3, 7.0, 0.1, 4, scheme=scheme, x_solver="chebyshev", **kw)
        for r in range(world):
            z0, nz = ret[r]["z"]
            np.testing.assert_allclose(ret[r]["ac_loss"], closs, rtol=1e-6)
            np.testing.assert_allclose(ret[r]["ac_x"], cx[z0:z0 + nz], rtol=2e-6, atol=2e-4)

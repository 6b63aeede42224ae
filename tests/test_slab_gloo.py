"""N > 1 path on CPU: world_size 2 and 3 ``gloo`` process groups exercise the z-slab partition, the
halo plan (which planes / which gradient channels travel, per scheme) and the scalar all-reduce of
pytv/slab.py.  The stencil arithmetic on each rank is done by the ORACLE on the halo-extended slab
(the HIP kernels need a GPU; their halo handling is checked bit-for-bit on one GPU in
test_gpu_parity.py::test_slab_calls_equal_unsharded), so a wrong plane, channel or direction in the
exchange shows up as a mismatch against the unsharded oracle result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, SCHEMES


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, lz, mu, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import tv_oracle as orc
        from pytv.slab import HaloPlan, Slab
        rng = np.random.default_rng(77)                       # same volume on every rank
        x_full = rng.standard_normal(shape)
        slab = Slab(shape[0])
        assert (slab.rank, slab.world) == (rank, world)
        assert sum(nz for _, nz in slab.parts) == shape[0] and slab.parts[rank] == (slab.z0, slab.nz)
        z0, nz = slab.z0, slab.nz
        errs = {}
        for scheme in SCHEMES:
            nd = orc.num_channels(scheme, shape[0], shape[1], lz, mu)
            y_full = np.random.default_rng(78).standard_normal((shape[0], nd) + shape[1:])
            plan = HaloPlan(slab, scheme, z_active=(shape[0] > 1 and lz > 0))
            x = torch.as_tensor(slab.local(x_full).copy())
            # ---- image halos -> D on the extended slab == rows of the unsharded D --------------------
            rp, rn = torch.full((1,) + shape[1:], np.nan, dtype=torch.float64), torch.full((1,) + shape[1:], np.nan, dtype=torch.float64)
            slab.wait(plan.exchange_image(x, rp, rn))
            lo = rp if plan.x_need_prev else torch.zeros_like(rp)     # planes the scheme never reads
            hi = rn if plan.x_need_next else torch.zeros_like(rn)
            ext = torch.cat([lo, x, hi]).numpy()
            # evaluate the oracle on a volume that has the TRUE planes wherever the scheme reads them
            probe = x_full.copy()
            probe[z0:z0 + nz] = ext[1:-1]
            if z0 > 0 and plan.x_need_prev:
                probe[z0 - 1] = ext[0]
            if z0 + nz < shape[0] and plan.x_need_next:
                probe[z0 + nz] = ext[-1]
            D_want = orc.D(x_full, scheme, lz, mu)[z0:z0 + nz]
            D_got = orc.D(probe, scheme, lz, mu)[z0:z0 + nz]
            assert not np.isnan(D_got).any(), scheme
            errs["D_" + scheme] = float(np.abs(D_got - D_want).max())
            # the received planes are exactly the neighbours' boundary planes
            if plan.x_need_prev:
                assert np.array_equal(rp[0].numpy(), x_full[z0 - 1]), scheme
            if plan.x_need_next:
                assert np.array_equal(rn[0].numpy(), x_full[z0 + nz]), scheme
            # ---- gradient halos -> D^T -----------------------------------------------------------------
            y = torch.as_tensor(y_full[z0:z0 + nz].copy())
            gp, gn = torch.full(shape[1:], np.nan, dtype=torch.float64), torch.full(shape[1:], np.nan, dtype=torch.float64)
            slab.wait(plan.exchange_grad(y, gp, gn))
            if plan.g_need_prev:
                assert np.array_equal(gp.numpy(), y_full[z0 - 1, plan.ch_back]), scheme
            if plan.g_need_next:
                assert np.array_equal(gn.numpy(), y_full[z0 + nz, plan.ch_fwd]), scheme
            probe_y = np.zeros_like(y_full)             # everything this rank does NOT hold is zero ...
            probe_y[z0:z0 + nz] = y.numpy()
            if plan.g_need_prev:
                probe_y[z0 - 1, plan.ch_back] = gp.numpy()       # ... except the received planes
            if plan.g_need_next:
                probe_y[z0 + nz, plan.ch_fwd] = gn.numpy()
            DT_want = orc.D_T(y_full, scheme, lz, mu)[z0:z0 + nz]
            DT_got = orc.D_T(probe_y, scheme, lz, mu)[z0:z0 + nz]
            errs["DT_" + scheme] = float(np.abs(DT_got - DT_want).max())
            # ---- two-plane halos for the radius-2 kernels ------------------------------------------------
            if min(n for _, n in slab.parts) >= 2:
                r2p = torch.full((2,) + shape[1:], np.nan, dtype=torch.float64) if slab.prev is not None else None
                r2n = torch.full((2,) + shape[1:], np.nan, dtype=torch.float64) if slab.next is not None else None
                slab.wait(plan.exchange_image2(x, r2p, r2n))
                if plan.on and r2p is not None:
                    assert np.array_equal(r2p.numpy(), x_full[z0 - 2:z0]), scheme
                if plan.on and r2n is not None:
                    assert np.array_equal(r2n.numpy(), x_full[z0 + nz:z0 + nz + 2]), scheme
            # ---- scalar all-reduce: TV of the slabs sums to the TV of the volume -------------------------
            part = torch.tensor([orc.compute_L21_norm(D_want)], dtype=torch.float64)
            slab.allreduce_sum_(part)
            total = orc.compute_L21_norm(orc.D(x_full, scheme, lz, mu))
            errs["tv_" + scheme] = abs(float(part[0]) - total) / total
        # ---- maximum over the ranks (the largest time weight of a sharded weight volume enters the CP step size) -----
        mx = torch.tensor([float(rank + 1), -float(rank)], dtype=torch.float64)
        slab.allreduce_max_(mx)
        assert mx.tolist() == [float(world), 0.0]
        # ---- the boundary planes of a per-voxel weight volume travel like image planes (solvers._SlabProblem) ----------
        W = np.random.default_rng(79).random(shape)
        wl = torch.as_tensor(slab.local(W).copy())
        gp, gn = torch.full((1,) + shape[1:], np.nan, dtype=torch.float64), torch.full((1,) + shape[1:], np.nan, dtype=torch.float64)
        slab.wait(slab.exchange(send_prev=wl[0:1] if slab.prev is not None else None, send_next=wl[nz - 1:nz] if slab.next is not None else None,
                                recv_prev=gp if slab.prev is not None else None, recv_next=gn if slab.next is not None else None))
        if slab.prev is not None:
            assert np.array_equal(gp[0].numpy(), W[z0 - 1])
        if slab.next is not None:
            assert np.array_equal(gn[0].numpy(), W[z0 + nz])
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,lz,mu", [(2, (6, 3, 5, 6), 1.3, 0.8), (3, (7, 2, 4, 5), 1.0, 1.0),
                                               (2, (4, 1, 6, 6), 0.0, 0.0)])
def test_halo_exchange_gloo(world, shape, lz, mu):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, lz, mu, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert v < 1e-12, (rank, k, v)


def test_partition_is_contiguous_and_balanced():
    sys.path.insert(0, PKG)
    from pytv.slab import partition
    for nz in (1, 7, 8, 64, 257):
        for w in (1, 2, 3, 8):
            if nz < w:
                continue
            parts = partition(nz, w)
            assert parts[0][0] == 0 and sum(n for _, n in parts) == nz
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1

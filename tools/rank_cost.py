#!/usr/bin/env python3
"""What ONE rank of an N-rank strong-scaling run computes per Chambolle-Pock iteration (north-star volume split into N z-slabs),
WITHOUT communication: the slab of a middle rank with its halo buffers left as they are (exchange and all-reduce are no-ops).
Gives the compute-only bound on the 1 -> N speed-up, T(1) / T_rank(N): what is left for halo cost and launch overhead before the
north star's ">= 6x at 8 GPUs" is missed.  No process group is needed.
usage: python tools/rank_cost.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from pytv.slab import Slab
from bench import synth_slab


class _Done:
    def wait(self):
        pass


class MuteSlab(Slab):
    """a Slab whose neighbour exchange and all-reduces do nothing (timing only: the halo planes keep whatever they hold)"""
    def exchange(self, send_prev=None, send_next=None, recv_prev=None, recv_next=None):
        return [_Done()]
    def allreduce_sum_(self, t):
        return t
    def allreduce_max_(self, t):
        return t


shape = (256, 8, 1024, 1024)
dev = torch.device("cuda", 0)
ns = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]
t1 = None
for n in ns:
    slab = MuteSlab(shape[0], rank=n // 2, world=n) if n > 1 else Slab(shape[0], rank=0, world=1)
    x0 = synth_slab(shape, slab.z0, slab.nz, dev)
    for overlap in ((True, False) if n > 1 else (True,)):
        cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, slab=slab, overlap=overlap)
        for _ in range(5):
            cp.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            cp.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
        if n == 1:
            t1 = dt
        print("N = %d: rank %d holds %3d planes, %s  %.3f ms per iteration (compute only)  -> speed-up bound %.2fx%s" % (
            n, slab.rank, slab.nz, "overlap schedule" if overlap else "plain schedule  ", dt, (t1 or dt) / dt,
            "" if t1 else "  (run with 1 first for the ratio)"), flush=True)
        del cp
    del x0
    torch.cuda.empty_cache()

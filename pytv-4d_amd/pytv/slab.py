"""z-slab sharding of an (Nz, M, N, N) volume over the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests).  The reference has no multi-GPU code at all; its layout remark
(README.md:235: "(Nz, M, N, N) ... can be decomposed easily along z") is what this module acts on:
rank r owns the contiguous planes [z0, z0 + nz) of both the image and the gradient/dual arrays, and
every operator apply needs ONE boundary plane (M*N*N elements, contiguous) from each neighbour.
Traffic is nearest-neighbour only -- a chain, not a ring: ncclSend/ncclRecv grouped in one
``batch_isend_irecv`` per exchange, plus an fp64 all-reduce of a few scalars for TV / loss / CG dots.
"""
import ctypes

import torch
import torch.distributed as dist


class _EventHandle:
    """work-handle look-alike: ``wait()`` orders the CURRENT stream behind an event (no host block)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class NativeComm:
    """The RCCL communicator of the C-ABI (include/pytv4d.h: tv_ctx_create / tv_halo_exchange / tv_allreduce_f64).

    ``torch.distributed`` is used ONCE, to hand rank 0's 128-byte unique id to the other ranks (any backend, gloo
    included); after that every exchange is a C call that enqueues ncclSend / ncclRecv on a dedicated side stream.
    The ordering with the launch stream is explicit -- an event recorded on the launch stream gates the side stream
    before the exchange, an event recorded after it is what ``wait()`` makes the launch stream wait for -- so nothing
    depends on ProcessGroupNCCL's internal streams."""

    def __init__(self, group=None, device=None):
        from . import _native as _nv
        self._nv = _nv
        self.lib = _nv.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _nv.check(self.lib.tv_ctx_unique_id(buf))
        box = [buf.raw]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        self.ctx = ctypes.c_void_p()
        _nv.check(self.lib.tv_ctx_create(ctypes.byref(self.ctx), self.rank, self.world, box[0], self.device.index or 0))
        self.stream = torch.cuda.Stream(self.device)

    def exchange(self, prev, nxt, send_prev, send_next, recv_prev, recv_next):
        ts = [t for t in (send_prev, send_next, recv_prev, recv_next) if t is not None]
        if not ts:
            return []
        count, dtype = ts[0].numel(), ts[0].dtype
        if any(t.numel() != count or t.dtype != dtype or not t.is_contiguous() for t in ts):
            raise ValueError("halo messages of one exchange must be contiguous and of one size / dtype")
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)                          # the planes to send are complete, the receive buffers' readers are done
        self.stream.wait_event(ready)
        p = self._nv.ptr
        self._nv.check(self.lib.tv_halo_exchange(self.ctx, self._nv.dtype_code(dtype), count, -1 if prev is None else prev,
                                                 -1 if nxt is None else nxt, p(send_prev), p(send_next), p(recv_prev), p(recv_next),
                                                 self.stream.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.stream)
        return [_EventHandle(done)]

    def allreduce_(self, t, op):
        if t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError("tv_allreduce_f64 reduces contiguous fp64 device words")
        self._nv.check(self.lib.tv_allreduce_f64(self.ctx, t.data_ptr(), t.numel(), op, torch.cuda.current_stream(self.device).cuda_stream))
        return t

    def close(self):
        if self.ctx:
            self.lib.tv_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()


def _span(t):
    """t itself if it is contiguous (or None), else the flat view of the storage span it covers"""
    if t is None or t.is_contiguous():
        return t
    n = 1 + sum((int(sz) - 1) * int(st) for sz, st in zip(t.shape, t.stride()))
    return t.as_strided((n,), (1,))


def partition(nz_global, world):
    """Contiguous, balanced split of nz_global planes: list of (z0, nz) per rank."""
    base, rem = divmod(int(nz_global), int(world))
    out, z0 = [], 0
    for r in range(world):
        nz = base + (1 if r < rem else 0)
        out.append((z0, nz))
        z0 += nz
    return out


class Slab:
    """This rank's share of the volume and its neighbour exchange."""

    def __init__(self, nz_global, group=None, rank=None, world=None, native_comm=None):
        """native_comm: a ``NativeComm`` -- halos and scalar all-reduces then go through the C-ABI's own RCCL
        communicator instead of ``torch.distributed`` (which the default path uses: backend "nccl" = RCCL)."""
        self.group = group
        self.native = native_comm
        if rank is None or world is None:
            if dist.is_available() and dist.is_initialized():
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world = 0, 1
        self.rank, self.world = int(rank), int(world)
        self.nz_global = int(nz_global)
        if self.nz_global < self.world:
            raise ValueError("cannot split %d planes over %d ranks" % (self.nz_global, self.world))
        parts = partition(self.nz_global, self.world)
        self.z0, self.nz = parts[self.rank]
        self.parts = parts
        self.prev = self.rank - 1 if self.rank > 0 else None
        self.next = self.rank + 1 if self.rank + 1 < self.world else None

    # ------------------------------------------------------------------------------------------
    @property
    def sharded(self):
        return self.world > 1

    def _global_rank(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def exchange(self, send_prev=None, send_next=None, recv_prev=None, recv_next=None):
        """Start one neighbour exchange.  Each argument is a contiguous tensor or None; sends/recvs
        towards a neighbour that does not exist are dropped.  Returns the list of work handles
        (``wait()`` orders the current stream behind the transfer on RCCL; it blocks on gloo)."""
        if not self.sharded:
            return []
        # planes of PITCHED arrays (tv_geom::row_pitch / frame_pitch) are strided views of one contiguous span of storage:
        # the span travels (its pads are zeros on both sides)
        send_prev, send_next, recv_prev, recv_next = (_span(t) for t in (send_prev, send_next, recv_prev, recv_next))
        if self.native is not None:
            return self.native.exchange(self.prev, self.next, send_prev if self.prev is not None else None,
                                        send_next if self.next is not None else None,
                                        recv_prev if self.prev is not None else None, recv_next if self.next is not None else None)
        if self._stage_through_host(send_prev, send_next, recv_prev, recv_next):
            return self._exchange_staged(send_prev, send_next, recv_prev, recv_next)
        ops = []
        if self.prev is not None:
            if recv_prev is not None:
                ops.append(dist.P2POp(dist.irecv, recv_prev, self._global_rank(self.prev), self.group))
            if send_prev is not None:
                ops.append(dist.P2POp(dist.isend, send_prev, self._global_rank(self.prev), self.group))
        if self.next is not None:
            if send_next is not None:
                ops.append(dist.P2POp(dist.isend, send_next, self._global_rank(self.next), self.group))
            if recv_next is not None:
                ops.append(dist.P2POp(dist.irecv, recv_next, self._global_rank(self.next), self.group))
        if not ops:
            return []
        return dist.batch_isend_irecv(ops)

    def _stage_through_host(self, *tensors):
        """gloo moves host memory only: device tensors are staged through the host.  This is the
        TEST configuration (several ranks sharing one GPU); production is RCCL, device to device."""
        if dist.get_backend(self.group) != "gloo":
            return False
        return any(t is not None and t.is_cuda for t in tensors)

    def _exchange_staged(self, send_prev, send_next, recv_prev, recv_next):
        def host(t):
            return None if t is None else t.detach().to("cpu").contiguous()

        def like(t):
            return None if t is None else torch.empty(t.shape, dtype=t.dtype, device="cpu")
        sp, sn, rp, rn = host(send_prev), host(send_next), like(recv_prev), like(recv_next)
        ops = []
        if self.prev is not None:
            if rp is not None:
                ops.append(dist.P2POp(dist.irecv, rp, self._global_rank(self.prev), self.group))
            if sp is not None:
                ops.append(dist.P2POp(dist.isend, sp, self._global_rank(self.prev), self.group))
        if self.next is not None:
            if sn is not None:
                ops.append(dist.P2POp(dist.isend, sn, self._global_rank(self.next), self.group))
            if rn is not None:
                ops.append(dist.P2POp(dist.irecv, rn, self._global_rank(self.next), self.group))
        for h in (dist.batch_isend_irecv(ops) if ops else []):
            h.wait()
        if self.prev is not None and rp is not None:
            recv_prev.copy_(rp)
        if self.next is not None and rn is not None:
            recv_next.copy_(rn)
        return []

    @staticmethod
    def wait(handles):
        for h in handles:
            h.wait()

    def allreduce_sum_(self, t):
        """In-place sum over ranks of a (small, fp64) tensor."""
        if self.sharded:
            if self.native is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous():
                return self.native.allreduce_(t, 0)
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.detach().to("cpu")
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_max_(self, t):
        """In-place maximum over ranks of a (small, fp64) tensor."""
        if self.sharded:
            if self.native is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous():
                return self.native.allreduce_(t, 1)
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.detach().to("cpu")
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def local(self, full):
        """This rank's planes of a full array whose axis 0 is z."""
        return full[self.z0:self.z0 + self.nz]


class HaloPlan:
    """Which boundary planes a scheme needs from / owes to its z-neighbours.

    Image halos (forward operator D):  upwind reads plane z0+nz (next), downwind plane z0-1 (prev),
    central and hybrid both.  Gradient halos (transposed operator): the adjoint of a forward
    difference looks BACKWARDS (needs the previous rank's last plane of the z channel), the adjoint
    of a backward difference looks FORWARDS; hybrid has one channel of each kind (z-up, z-down),
    central reads the same channel on both sides.  A rank sends the mirror image of what its
    neighbour needs.  Only the z channel plane(s) travel: M*N*N elements per message."""

    def __init__(self, slab, scheme, z_active):
        self.slab = slab
        on = bool(slab.sharded and z_active)
        has_p, has_n = slab.prev is not None, slab.next is not None
        self.x_need_prev = on and scheme != "upwind" and has_p
        self.x_need_next = on and scheme != "downwind" and has_n
        self.x_send_prev = on and scheme != "downwind" and has_p     # my first plane is prev's "next"
        self.x_send_next = on and scheme != "upwind" and has_n       # my last plane is next's "prev"
        self.g_need_prev = on and scheme != "downwind" and has_p
        self.g_need_next = on and scheme != "upwind" and has_n
        self.g_send_next = on and scheme != "downwind" and has_n     # my last backward-looking plane
        self.g_send_prev = on and scheme != "upwind" and has_p       # my first forward-looking plane
        per = 2 if scheme == "hybrid" else 1
        self.ch_back = 2 * per                                       # z channel whose adjoint looks backwards
        self.ch_fwd = 2 * per + (1 if scheme == "hybrid" else 0)     # z channel whose adjoint looks forwards
        self.on = on

    def exchange_image(self, x, recv_prev, recv_next):
        """One-plane image halos: x is (nz, M, N, N); recv_* are (1, M, N, N) buffers or None."""
        nz = x.shape[0]
        return self.slab.exchange(send_prev=x[0:1] if self.x_send_prev else None,
                                  send_next=x[nz - 1:nz] if self.x_send_next else None,
                                  recv_prev=recv_prev if self.x_need_prev else None,
                                  recv_next=recv_next if self.x_need_next else None)

    def exchange_image2(self, x, recv_prev, recv_next):
        """Two-plane image halos for the radius-2 kernels (sub-gradient, normal operator)."""
        if not self.on:
            return []
        nz = x.shape[0]
        return self.slab.exchange(send_prev=x[0:2], send_next=x[nz - 2:nz], recv_prev=recv_prev, recv_next=recv_next)

    def exchange_grad(self, q, recv_prev, recv_next):
        """Gradient halos: q is (nz, Nd, M, N, N); recv_* are (M, N, N)-shaped buffers or None."""
        nz = q.shape[0]
        return self.slab.exchange(send_prev=q[0, self.ch_fwd] if self.g_send_prev else None,
                                  send_next=q[nz - 1, self.ch_back] if self.g_send_next else None,
                                  recv_prev=recv_prev if self.g_need_prev else None,
                                  recv_next=recv_next if self.g_need_next else None)

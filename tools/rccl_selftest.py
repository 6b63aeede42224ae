#!/usr/bin/env python3
"""One-rank RCCL check of the stream hand-off pytv/slab.py relies on (run by tests/test_gpu_rccl.py in a fresh process;
needs RANK=0 WORLD_SIZE=1 MASTER_ADDR MASTER_PORT).

With two ranks, rank r sends its first plane to r-1 and receives r+1's first plane as its "next" halo.  Here the one
rank plays both: the volume is cut in two slabs A = planes [0, h) and B = planes [h, nz); B's first plane travels
through ``batch_isend_irecv`` (RCCL's own stream) into A's halo buffer and tv_D of slab A runs right after
``work.wait()``.  The exchange is repeated while the halo buffer's previous consumer and an unrelated writer of OTHER
planes are still queued, which is the interior-first schedule of the solvers.  Everything is compared with the
unsharded tv_D at the end -- the only host synchronisation of the script."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from pytv import _native as nv
    lib = nv.lib()
    shape = (8, 4, 256, 512)
    nz, h = shape[0], 4
    gen = torch.Generator(device=dev).manual_seed(1)
    ok = True
    for scheme in ("upwind", "hybrid"):
        gF = nv.Geometry(shape, scheme, torch.float32, dev, reg_time=1.0)
        gA = nv.Geometry((h,) + shape[1:], scheme, torch.float32, dev, reg_time=1.0, nz_global=nz, z0=0)
        x = torch.empty(shape, device=dev)
        halo = torch.zeros((1,) + shape[1:], device=dev)
        dA = torch.empty(gA.grad_shape, device=dev)
        dF = torch.empty(gF.grad_shape, device=dev)
        results = []
        for rep in range(6):
            x.copy_(torch.rand(shape, device=dev, generator=gen) * 100)           # producer kernel on the launch stream
            ops = [dist.P2POp(dist.irecv, halo, 0), dist.P2POp(dist.isend, x[h:h + 1], 0)]
            works = dist.batch_isend_irecv(ops)
            # interior work that neither reads the halo nor writes the plane being sent, queued while the transfer runs
            nv.check(lib.tv_D(gF.ref, nv.ptr(x), None, None, nv.ptr(dF), nv.current_stream(dev)))
            for w in works:
                w.wait()                                                           # launch stream waits for RCCL's stream
            nv.check(lib.tv_D(gA.ref, nv.ptr(x[0:h]), None, nv.ptr(halo), nv.ptr(dA), nv.current_stream(dev)))
            results.append((dA.clone(), dF[0:h].clone()))
        torch.cuda.synchronize()
        for rep, (a, f) in enumerate(results):
            same = torch.equal(a, f)
            ok = ok and same
            print("scheme %s rep %d: slab A with the received halo == unsharded: %s" % (scheme, rep, same), flush=True)
    t = torch.ones(3, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    ok = ok and bool((t == 1).all().item())
    dist.destroy_process_group()
    print("RCCL_SELFTEST_OK" if ok else "RCCL_SELFTEST_FAILED", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

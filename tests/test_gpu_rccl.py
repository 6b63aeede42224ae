"""First contact with RCCL on a real MI355X.  A 1-GPU box cannot host two RCCL ranks, but everything else of the
production multi-GPU path can run with a communicator of ONE rank: ``init_process_group("nccl", device_id=...)``,
``dist.barrier``, the fp64 device ``all_reduce`` of the loss history and of the timing, ``destroy_process_group`` --
the calls bench.py makes at N = 8 -- and a device-to-device ``batch_isend_irecv`` on the communicator's own stream with
the current stream ordered behind it by ``work.wait()`` (the hand-off pytv/slab.py relies on).
Each check runs in a FRESH child process (a process that has touched the GPU is never re-exec'ed)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env(port):
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TV_BENCH_BACKEND", None)
    env.pop("TV_BENCH_SHARE_GPU", None)
    return env


def test_bench_on_a_one_rank_rccl_communicator():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "small", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline"], env=_env(29631), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["rccl_ranks"] == 1 and out["halo"]["backend"] == "nccl"
    assert out["n_gpus"] == 1 and out["value"] > 0
    first, last = out["loss_first_last"]
    assert last < first


def test_self_exchange_on_the_rccl_stream_is_ordered_with_the_launch_stream():
    """tools/rccl_selftest.py: a plane produced by a kernel is sent (to this same rank) while an unrelated kernel runs,
    received into a halo buffer that an earlier kernel was still reading, and consumed by a HIP kernel (tv_D with that
    halo) right after work.wait() -- no host synchronisation anywhere; the result must equal the unsharded tv_D."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selftest.py")], env=_env(29632), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "RCCL_SELFTEST_OK" in p.stdout

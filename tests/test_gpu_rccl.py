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
# small test volumes take the one-sweep Chambolle-Pock path too (the other GPU test modules set this at import time; this module's
# bench.py children inherit it -- set here as well so that the module also passes when it is run on its own)
os.environ["TV_FUSED_MIN_KVOXELS"] = "0"


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


_CTX_SCRIPT = r"""
import ctypes, sys
sys.path.insert(0, %r)
import torch
from pytv import _native as nv
lib = nv.lib()
torch.cuda.set_device(0)
buf = ctypes.create_string_buffer(128)
nv.check(lib.tv_ctx_unique_id(buf))
ctx = ctypes.c_void_p()
nv.check(lib.tv_ctx_create(ctypes.byref(ctx), 0, 1, buf.raw, 0))
assert lib.tv_ctx_rank(ctx) == 0 and lib.tv_ctx_size(ctx) == 1
st = torch.cuda.current_stream().cuda_stream
for dtype, code in ((torch.float32, 0), (torch.float64, 1)):
    a = torch.rand(1 << 20, device="cuda", dtype=dtype)          # what a rank sends towards its previous neighbour
    b = torch.rand(1 << 20, device="cuda", dtype=dtype)          # ... towards its next neighbour
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    # the one rank is its own prev and next peer: messages meet in posting order (send_prev -> recv_prev, send_next -> recv_next)
    nv.check(lib.tv_halo_exchange(ctx, code, a.numel(), 0, 0, a.data_ptr(), b.data_ptr(), ra.data_ptr(), rb.data_ptr(), st))
    c = ra * 2 + rb                                               # consumer on the same stream, no host sync in between
    torch.cuda.synchronize()
    assert torch.equal(ra, a) and torch.equal(rb, b) and torch.equal(c, a * 2 + b)
    # a rank at the end of the chain: only one neighbour
    rb.zero_()
    nv.check(lib.tv_halo_exchange(ctx, code, a.numel(), -1, 0, None, b.data_ptr(), None, rb.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(rb, b)
t = torch.tensor([1.5, -2.0, 7.0], dtype=torch.float64, device="cuda")
nv.check(lib.tv_allreduce_f64(ctx, t.data_ptr(), 3, 0, st))
nv.check(lib.tv_allreduce_f64(ctx, t.data_ptr(), 3, 1, st))
torch.cuda.synchronize()
assert t.tolist() == [1.5, -2.0, 7.0]
assert lib.tv_halo_exchange(ctx, 0, 4, 5, -1, None, None, None, None, st) < 0        # neighbour outside the communicator
nv.check(lib.tv_ctx_destroy(ctx))
print("TV_CTX_OK")
"""


def test_cabi_rccl_context_with_one_rank():
    """include/pytv4d.h multi-GPU surface (tv_ctx_create / tv_halo_exchange / tv_allreduce_f64 / tv_ctx_destroy) on a real
    RCCL communicator of one rank, the rank being its own z-neighbour."""
    pkg = os.path.join(ROOT, "pytv-4d_amd")
    p = subprocess.run([sys.executable, "-c", _CTX_SCRIPT % pkg], env=_env(29633), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "TV_CTX_OK" in p.stdout


def _bench_line(cmd, env):
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_sharded_over_ranks_gives_the_single_rank_loss(ranks):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with the TEST
    transport (gloo, the ranks sharing cuda:0, host-staged halos): the N-rank job works on the SAME volume split in
    z-slabs, so its loss history must equal the single-process one; rank 0 prints one line with n_gpus = N and the CPU
    baseline.  What this cannot cover is RCCL with N > 1 (needs N GPUs)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TV_BENCH_BACKEND", "TV_BENCH_SHARE_GPU"):
        env.pop(k, None)
    common = ["--workload", "small", "--steps", "6", "--warmup", "2"]
    one = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--pmc", "off"] + common, env)
    env2 = dict(env, TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_ZCHUNK="1")      # short chunks: interior-first schedule
    many = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(29640 + ranks), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(ranks)] + common, env2)
    assert many["n_gpus"] == ranks and many["steps"] == 6 and many["scaling"] == "strong"
    assert "cpu_baseline" in many and many["cpu_baseline"]["cores"] == 1
    assert "roofline" in many and many["rccl_ranks"] == 0            # the test transport is not RCCL
    a, b = one["loss_first_last"], many["loss_first_last"]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-6 * abs(a[1])
    # N > 1: the line says which transport ran and where an iteration's time went (max / mean over ranks per phase)
    assert "gloo" in many["comm"] and "TEST" in many["comm"] and "none" in one["comm"]
    ph = many["phases"]
    want = {"sweep_interior_a", "x_halo_wait_exposed", "sweep_edges", "sweep_interior_b", "fixup_interior", "q_halo_wait_exposed", "fixup_edges"}
    assert set(ph["max_over_ranks"]) == want == set(ph["mean_over_ranks"]) == set(ph["rank0"])
    assert all(v >= 0.0 for v in ph["max_over_ranks"].values())
    assert all(ph["max_over_ranks"][k] >= ph["mean_over_ranks"][k] - 1e-9 for k in want)
    assert sum(ph["mean_over_ranks"].values()) <= 1.5 * many["ms_per_step"]
    assert "phases" not in one


def test_n8_rehearsal_eight_ranks_on_one_gpu():
    """Round-3 verdict item 8: no 8-GPU box is available to the build, so the N = 8 job is rehearsed with everything but RCCL --
    `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` exactly as the driver launches it (the launcher starts
    BEFORE anything touches the GPU), eight ranks sharing cuda:0 over the TEST transport (gloo, host-staged halos), on the
    north-star frame (64 x 8 x 1024 x 1024: 8 planes per rank, 32 MiB halo planes).  The 8-rank loss must equal the 1-rank loss,
    all seven phase fields of the interior-first schedule must be present, the line must say n_gpus = 8 / strong scaling.
    This is NOT a scaling measurement (none exists)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TV_BENCH_BACKEND", "TV_BENCH_SHARE_GPU"):
        env.pop(k, None)
    common = ["--workload", "rehearsal", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--pmc", "off"]
    env1 = dict(env, TV_ZCHUNK="2")
    one = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env1)
    env8 = dict(env, TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_ZCHUNK="2")      # 2-plane chunks: 4 chunks per 8-plane slab
    many = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29658", os.path.join(ROOT, "bench.py"), "--gpus", "8"] + common, env8)
    assert many["n_gpus"] == 8 and many["steps"] == 4 and many["scaling"] == "strong" and many["rccl_ranks"] == 0
    assert many["config"]["shape"] == [64, 8, 1024, 1024] and "z-slab x8" in many["config"]["parallelism"]
    a, b = one["loss_first_last"], many["loss_first_last"]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-6 * abs(a[1])
    ph = many["phases"]
    want = {"sweep_interior_a", "x_halo_wait_exposed", "sweep_edges", "sweep_interior_b", "fixup_interior", "q_halo_wait_exposed", "fixup_edges"}
    assert set(ph["max_over_ranks"]) == want == set(ph["mean_over_ranks"]) == set(ph["rank0"])
    assert many["halo"]["bytes_per_plane"] == 32 * 1024 * 1024 and many["halo"]["exchanges_per_iteration"] == 2
    assert "series_ms" in many and "gpu_state" in many


def test_bench_admm_line_on_one_gpu():
    """Round-4 verdict item 3: `python bench.py --solver admm` prints the same JSON shape as the CP bench -- metric
    admm_outer_iters_per_sec, a roofline block on the one-sweep dual-side kernel with its words per voxel, the x-solve beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "admm-small", "--steps", "6", "--warmup", "2",
                       "--pmc", "off", "--scheme", "upwind"], env)
    assert out["metric"] == "admm_outer_iters_per_sec" and out["unit"] == "it/s" and out["value"] > 0 and out["n_gpus"] == 1
    assert out["config"]["workload"].startswith("admm-small 16x4x256x256") and out["config"]["x_solver"] == "chebyshev"
    nd, n_cg = out["config"]["nd"], out["config"]["n_cg"]
    w = out["words_per_voxel_and_outer_iteration"]
    assert w["sweep"] == 2 * nd + 3 and w["total"] == w["sweep"] + w["xsolve"] and 0 < w["xsolve"] <= 4 * n_cg - 6
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["bytes_per_launch"] == 4.0 * w["sweep"] * 16 * 4 * 256 * 256
    assert 0 < out["roofline_xsolve"]["frac"] < 1 and out["roofline_xsolve"]["launches_per_outer_iteration"] >= 1
    first, last = out["loss_first_last"]
    assert last < first
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0


@pytest.mark.parametrize("workload,nz,extra,port", [("config3", 64, [], 29661), ("config4", 32, ["--scheme", "upwind"], 29662),
                                                    ("config4", 32, ["--scheme", "hybrid"], 29663)])
def test_n8_rehearsal_of_the_two_8gpu_configs(workload, nz, extra, port):
    """BASELINE configs[3] (512 x 8 x 1024 x 1024 hybrid CP) and configs[4] (256 x 16 x 1024 x 1024 ADMM) as the driver would launch
    them at N = 8, with the plane count scaled down (--nz) so that eight TEST ranks (gloo, host-staged halos) fit ONE GPU: the 8-rank
    loss must equal the 1-rank loss of the same scaled volume.  Not a scaling measurement."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TV_BENCH_BACKEND", "TV_BENCH_SHARE_GPU"):
        env.pop(k, None)
    common = ["--workload", workload, "--nz", str(nz), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc", "off"] + extra
    one = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--allow-single", "--tune-placement", "off"] + common,
                      dict(env, TV_ZCHUNK="2"))
    env8 = dict(env, TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_ZCHUNK="2")
    many = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8",
                        "--tune-placement", "off"] + common, env8)
    assert many["n_gpus"] == 8 and many["scaling"] == "strong" and many["config"]["shape"][0] == nz
    assert many["metric"] == one["metric"] == ("admm_outer_iters_per_sec" if workload == "config4" else "chambolle_pock_iters_per_sec")
    a, b = one["loss_first_last"], many["loss_first_last"]
    assert abs(a[0] - b[0]) <= 2e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 2e-6 * abs(a[1]), (a, b)
    assert "roofline" in many and "z-slab x8" in many["config"]["parallelism"]


def test_bench_reports_a_failed_communicator_setup_and_exits_nonzero():
    """round-2 verdict item 3c: if the communicator cannot be set up, rank 0 prints a JSON line carrying the error and the
    process exits non-zero (fresh process; nothing is re-exec'ed).  Provoked with a backend name that does not exist."""
    env = _env(29633)
    env["TV_BENCH_BACKEND"] = "no_such_backend"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "small", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--pmc", "off"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and "communicator setup failed" in out["error"] and out["n_gpus"] == 1


def test_bench_error_path_under_the_launcher_exits_nonzero():
    """the same failure with several ranks under torch.distributed.run (how the driver starts N > 1): the launcher's exit code is
    non-zero and rank 0's JSON line names the error"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(TV_BENCH_BACKEND="no_such_backend", TV_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29659", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--pmc", "off"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-1500:]
    out = json.loads(lines[-1])
    assert out["value"] is None and "communicator setup failed" in out["error"] and out["n_gpus"] == 2


def test_bench_measures_its_hbm_traffic_live():
    """`python bench.py` on one GPU (no launcher environment): roofline.traffic comes from two rocprofv3 --pmc child passes of the
    same command, not from the committed table; it can only exceed the algorithmic bytes (by the halo rows / prologues)."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    r = out["roofline"]
    assert r["traffic_source"].startswith("LIVE"), r["traffic_source"]
    assert 0.98 * r["bytes_per_launch"] <= r["traffic"] <= 1.25 * r["bytes_per_launch"], (r["traffic"], r["bytes_per_launch"])
    assert out["roofline_fixup"]["traffic"] > 0


def _shell_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TV_BENCH_BACKEND", "TV_BENCH_SHARE_GPU"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_gpus_2_from_a_plain_shell_launches_itself_and_matches_one_rank():
    """Round-5 verdict item 1: `python3 bench.py --gpus 2 --workload rehearsal --nz 8` from a plain shell (no launcher, no RANK /
    WORLD_SIZE) starts torch.distributed.run as a child before anything touches the GPU, relays rank 0's line and exit code; with the
    TEST transport (gloo, both ranks on cuda:0) the 2-rank loss equals the 1-rank loss."""
    common = ["--workload", "rehearsal", "--nz", "8", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--pmc", "off"]
    one = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, _shell_env(TV_ZCHUNK="2"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                       env=_shell_env(TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_ZCHUNK="2"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "self-launch (torch imported: False)" in p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["shape"] == [8, 8, 1024, 1024] and "z-slab x2" in two["config"]["parallelism"]
    assert two["rccl_ranks"] == 0 and "phases" in two and "x_halo_wait_exposed" in two["phases"]["max_over_ranks"]
    a, b = one["loss_first_last"], two["loss_first_last"]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-6 * abs(a[1])
    assert one["n_gpus"] == 1 and "self-launch" not in one.get("comm", "")


def test_a_stalled_rank_ends_the_job_with_an_error_line_not_a_hang():
    """the per-phase watchdog: rank 1 stops before the timed region (TEST knob), rank 0 sits in the barrier; every rank's deadline
    passes, rank 0 prints the error line, all processes exit non-zero and the self-launching parent relays line and code."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--pmc", "off"],
                       env=_shell_env(TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_BENCH_TEST_STALL="1:timed", TV_BENCH_WATCHDOG_SCALE="0.15"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "watchdog" in out["error"] and "timed region" in out["error"]


def test_comm_cabi_through_the_self_launch_reports_its_setup_failure():
    """--comm cabi needs one GPU per rank (RCCL refuses two ranks on one device): from a plain shell on this 1-GPU box the 2-rank job
    must come back with the communicator-setup error line and a non-zero exit code -- relayed, not hung."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--pmc", "off", "--comm", "cabi"],
                       env=_shell_env(TV_BENCH_BACKEND="gloo", TV_BENCH_SHARE_GPU="1", TV_BENCH_WATCHDOG_SCALE="0.25"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and out["n_gpus"] == 2
    assert "communicator setup failed" in out["error"] or "watchdog" in out["error"]

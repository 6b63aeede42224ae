"""CPU-side checks of bench.py's command line (nothing here touches a GPU): the BASELINE configs it knows, the refusal of the
8-GPU workloads on one rank, the bounded CPU-baseline samples, and the word counts of the ADMM line against the solver's own."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_workloads_cover_the_baseline_configs():
    w = bench.WORKLOADS
    assert w["northstar"]["shape"] == (256, 8, 1024, 1024)
    assert w["config1"]["shape"] == (256, 1, 512, 512) and w["config2"]["shape"] == (128, 8, 512, 512)
    assert w["config3"]["shape"] == (512, 8, 1024, 1024) and w["config3"]["min_gpus"] >= 2
    assert w["config4"]["shape"] == (256, 16, 1024, 1024) and w["config4"]["solver"] == "admm" and w["config4"]["min_gpus"] >= 2
    # what one rank of the two 8-GPU jobs holds
    assert w["config3-slab"]["shape"] == (512 // 8, 8, 1024, 1024)
    assert w["config4-slab"]["shape"] == (256 // 8, 16, 1024, 1024) and w["config4-slab"]["solver"] == "admm"
    assert set(bench.METRIC) == {"cp", "admm"}


@pytest.mark.parametrize("workload,metric", [("config3", "chambolle_pock_iters_per_sec"), ("config4", "admm_outer_iters_per_sec")])
def test_multi_gpu_workloads_are_refused_on_one_rank_before_the_gpu_is_touched(workload, metric):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and out["metric"] == metric and "--allow-single" in out["error"] and out["n_gpus"] == 1


def test_cpu_baseline_samples_are_bounded_and_say_what_they_are(monkeypatch):
    """The samples are sized by voxel count, not by the workload: shrink the targets so that the test takes a second, then check
    the bookkeeping (4 planes for CP as the survey asks, extrapolation linear in the voxel count)."""
    from oracle import tv_oracle as orc
    calls = {}

    def fake_cp(x0, n_it, reg, **kw):
        calls["cp"] = (x0.shape, n_it, kw["scheme"])
        return x0, np.zeros(n_it)

    def fake_admm(x0, n_outer, reg, rho, n_cg, **kw):
        calls["admm"] = (x0.shape, n_outer, n_cg, kw["scheme"], kw["x_solver"])
        return x0, np.zeros(n_outer)

    monkeypatch.setattr(orc, "chambolle_pock", fake_cp)
    monkeypatch.setattr(orc, "admm", fake_admm)
    b = bench.cpu_baseline((256, 8, 1024, 1024), 1.0, 1.0, 8, "hybrid")
    assert calls["cp"][0] == (4, 8, 1024, 1024) and calls["cp"][2] == "hybrid"          # V ~ 3e7, e.g. 4 planes
    assert b["cores"] == 1 and b["kind"] == "port" and "(4, 8, 1024, 1024)" in b["sample"]
    a = bench.cpu_baseline_admm((32, 16, 1024, 1024), 1.0, 1.0, "upwind", 0.05, 5)
    shp = calls["admm"][0]
    assert shp[0] == 2 and shp[1] == 16 and shp[3] == 1024 and shp[2] % 16 == 0 and 1.0e7 <= np.prod(shp) <= 2.0e7
    assert calls["admm"][1:] == (1, 5, "upwind", "chebyshev")
    assert a["unit"] == "outer it/s" and a["cores"] == 1 and "sub-volume" in a["sample"]
    h = bench.cpu_baseline_admm((32, 16, 1024, 1024), 1.0, 1.0, "hybrid", 0.05, 5)
    assert np.prod(calls["admm"][0]) < np.prod(shp)                                      # hybrid is 3 x slower per voxel: smaller sample
    assert h["value"] > 0


def _plain_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env["TV_BENCH_LAUNCH_MODULE"] = "_record_launch"
    env.update(extra)
    return env


def test_gpus_n_from_a_plain_shell_starts_the_launcher_as_a_child_before_torch_is_imported():
    """Round-5 verdict item 1: `python3 bench.py --gpus 8` without RANK / WORLD_SIZE used to exit 1.  Now the process starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py <same arguments>` as a child, relays its stdout and exit
    code, and never imports torch itself (here the launcher module is replaced by a recorder)."""
    argv = ["--gpus", "8", "--steps", "7", "--warmup", "2", "--workload", "config3", "--comm", "cabi"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_plain_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1500:]
    assert "self-launch (torch imported: False)" in p.stderr
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    a = rec["recorded_argv"]
    assert a[:3] == ["--nnodes=1", "--nproc-per-node", "8"] and a[3:5] == ["--master-addr", "127.0.0.1"] and a[5] == "--master-port"
    assert int(a[6]) > 0 and os.path.samefile(a[7], os.path.join(ROOT, "bench.py")) and a[8:] == argv
    assert rec["self_launched"] == "1" and rec["ipc_legacy"] == "0"
    # (config3 on 8 ranks is NOT refused by the parent: the min_gpus rule is the ranks' to apply, with WORLD_SIZE set)


def test_self_launch_relays_the_exit_code_and_always_leaves_a_line():
    argv = ["--gpus", "2", "--workload", "small"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_plain_env(TV_FAKE_LAUNCH_RC="7", TV_BENCH_RETRY="0"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 7
    # a launcher that dies without a line: the parent prints the error line itself
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_plain_env(TV_FAKE_LAUNCH_RC="9", TV_FAKE_LAUNCH_SILENT="1", TV_BENCH_RETRY="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 9
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "printed no line" in out["error"]
    # a launcher that hangs: killed by process group after TV_BENCH_TIMEOUT, error line, exit code 5
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv,
                       env=_plain_env(TV_FAKE_LAUNCH_SILENT="1", TV_FAKE_LAUNCH_SLEEP="600", TV_BENCH_TIMEOUT="3"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 5
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and "did not finish within" in out["error"]
    # a failed attempt is repeated: without the overlap, then with host-staged halos over gloo (TV_BENCH_RETRY=0: one attempt only)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_plain_env(TV_FAKE_LAUNCH_RC="3"),
                       capture_output=True, text=True, timeout=120)
    recs = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 3 and len(recs) == 3 and "--no-overlap" not in recs[0]["recorded_argv"]
    assert recs[1]["recorded_argv"][-1] == "--no-overlap" and recs[1]["backend"] is None
    assert recs[2]["recorded_argv"][-1] == "--no-overlap" and recs[2]["backend"] == "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_plain_env(TV_FAKE_LAUNCH_RC="3", TV_BENCH_RETRY="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1


def test_under_a_launcher_bench_does_not_launch_again():
    """with RANK / WORLD_SIZE in the environment (torch.distributed.run, the driver's form) main() goes straight on: the refusal of a
    multi-GPU workload on too few ranks is the first thing that can be observed without a GPU"""
    env = _plain_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "config3"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "self-launch" not in p.stderr
    assert "multi-GPU job" in json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["error"]


def test_watchdog_ends_a_stuck_process_with_an_error_line():
    """bench.Watchdog alone (no GPU): an armed deadline that passes prints rank 0's error line and ends the process with code 4."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench, argparse\n"
            "a = argparse.Namespace(solver='cp', steps=3, warmup=1, workload='small')\n"
            "w = bench.Watchdog(a, 0, 2); w.arm('first', 600); w.arm('timed region', 1.0); time.sleep(30); print('NOT REACHED')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 4 and "NOT REACHED" not in p.stdout
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["value"] is None and out["n_gpus"] == 2 and "timed region" in out["error"] and "rank 0 of 2" in out["error"]
    code2 = code.replace("time.sleep(30)", "w.disarm(); time.sleep(3)")
    p = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "NOT REACHED" in p.stdout

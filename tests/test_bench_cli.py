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

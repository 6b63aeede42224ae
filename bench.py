#!/usr/bin/env python3
"""Benchmark of the hot path: fused Chambolle-Pock iterations (default) or ADMM outer iterations (--solver admm) on a
synthetic (Nz, M, N, N) fp32 volume.

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or from a plain shell -- bench.py then
                                                             starts the launcher itself as a child, before anything touches a GPU)
    python bench.py --solver admm --workload config4-slab --scheme upwind      (BASELINE configs[4]: the per-GPU slab of the 8-GPU job)

A "step" is ONE Chambolle-Pock iteration (README.md:145-157 of the reference: fidelity-dual update,
D + prox dual update, D^T primal update, loss) over the whole volume, state resident in HBM.  With
N > 1 the SAME volume is split into N contiguous z-slabs (strong scaling), one process per GPU, one
boundary plane per neighbour exchanged over RCCL/xGMI before each of the two kernels and hidden behind
the interior planes.  Rank 0 prints ONE JSON line.

Workloads (--workload):
  northstar  (256, 8, 1024, 1024) hybrid, reg_z = reg_time = 1   <- default; the shape BASELINE.json's
             60 %-of-HBM-peak target is quoted on (fits one GPU: x,x0,p 24 GiB + q 64 GiB)
  config1    (256, 1, 512, 512)   3-D hybrid        (BASELINE.json configs[1])
  config2    (128, 8, 512, 512)   4-D hybrid        (BASELINE.json configs[2])
  small      (16, 4, 256, 256)    quick functional run
  rehearsal  (64, 8, 1024, 1024)  the north-star frame, 64 planes: 8 test ranks sharing one GPU (tests/test_gpu_rccl.py);
             also the per-GPU slab of config3 at N = 8 ("config3-slab" is the same shape)
  config3    (512, 8, 1024, 1024) 4-D hybrid CP, the z-slab job of BASELINE.json configs[3] (128 GiB of q: meant for N = 8, 64 planes
             per rank; N = 1 only with --allow-single -- 192 GiB of state, placement tuner off)
  config4    (256, 16, 1024, 1024) ADMM, all four schemes via --scheme (BASELINE.json configs[4]; N = 1 only with --allow-single and
             only for the Nd = 4 schemes: hybrid needs 352 GiB)
  config4-slab (32, 16, 1024, 1024) what ONE rank of config4 holds at N = 8: the single-GPU line of the ADMM path
  admm-small (16, 4, 256, 256)    quick functional ADMM run

--solver cp | admm: default = the solver the workload names (config4* / admm-small: admm, everything else: cp).  A "step" of the ADMM
bench is ONE outer iteration of pytv.solvers.ADMM as it comes by default (one-sweep dual side tv_admm_sweep + tv_admm_fixup,
Chebyshev x-solve with --n-cg steps, keep_z=True); metric admm_outer_iters_per_sec.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytv-4d_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    "northstar": dict(shape=(256, 8, 1024, 1024), reg_z=1.0, reg_time=1.0),
    "config1": dict(shape=(256, 1, 512, 512), reg_z=1.0, reg_time=0.0),
    "config2": dict(shape=(128, 8, 512, 512), reg_z=1.0, reg_time=1.0),
    "small": dict(shape=(16, 4, 256, 256), reg_z=1.0, reg_time=1.0),
    # the north-star frame (8 x 1024 x 1024) with 64 planes: what 8 ranks sharing ONE GPU can hold -- the N = 8 rehearsal of
    # tests/test_gpu_rccl.py (no 8-GPU box is available to the build; the driver runs the real curve)
    "rehearsal": dict(shape=(64, 8, 1024, 1024), reg_z=1.0, reg_time=1.0),
    "config3-slab": dict(shape=(64, 8, 1024, 1024), reg_z=1.0, reg_time=1.0),
    # BASELINE.json configs[3] / configs[4]: the two 8-GPU jobs.  min_gpus: fewer ranks are refused unless --allow-single
    "config3": dict(shape=(512, 8, 1024, 1024), reg_z=1.0, reg_time=1.0, min_gpus=2),
    "config4": dict(shape=(256, 16, 1024, 1024), reg_z=1.0, reg_time=1.0, min_gpus=2, solver="admm"),
    "config4-slab": dict(shape=(32, 16, 1024, 1024), reg_z=1.0, reg_time=1.0, solver="admm"),
    "admm-small": dict(shape=(16, 4, 256, 256), reg_z=1.0, reg_time=1.0, solver="admm"),
}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
METRIC = {"cp": "chambolle_pock_iters_per_sec", "admm": "admm_outer_iters_per_sec"}


def synth_slab(shape, z0, nz, device, seed=1234):
    """Noisy piecewise-constant phantom, generated per global plane so that any slab of
    it is identical no matter how many ranks share the volume."""
    import torch
    Nz, M, Ny, Nx = shape
    x = torch.zeros((nz, M, Ny, Nx), dtype=torch.float32, device=device)
    rng = np.random.RandomState(seed)
    for _ in range(32):
        c = rng.rand(3)
        h = 0.05 + 0.25 * rng.rand(3)
        amp = float(rng.rand() * 255.0 / 4.0)
        za, zb = int(np.floor((c[0] - h[0]) * Nz)), int(np.ceil((c[0] + h[0]) * Nz))
        ya, yb = max(0, int((c[1] - h[1]) * Ny)), min(Ny, int((c[1] + h[1]) * Ny))
        la, lb = max(za, z0) - z0, min(zb, z0 + nz) - z0
        if lb <= la or yb <= ya:
            continue
        for t in range(M):
            xa, xb = max(0, int((c[2] - h[2]) * Nx) + t), min(Nx, int((c[2] + h[2]) * Nx) + t)
            if xb > xa:
                x[la:lb, t, ya:yb, xa:xb] += amp
    gen = torch.Generator(device=device)
    for k in range(nz):
        gen.manual_seed(seed * 100003 + z0 + k)
        x[k] += 100.0 * torch.rand((M, Ny, Nx), dtype=torch.float32, device=device, generator=gen)
    return x


def cpu_baseline(shape, reg_z, reg_time, nd, scheme="hybrid"):
    """The NumPy oracle (a single-threaded restatement of pytv.tv_CPU / tv_operators_CPU, pinned to
    the reference by tests/golden) timed on a bounded z-sub-slab of the same workload."""
    from oracle import tv_oracle as orc
    Nz, M, Ny, Nx = shape
    # V ~ 3e7 voxels (round-4 verdict: 2 planes made half of the z differences boundary zeros): 4 planes of the north-star
    # frame, ~11 s per iteration of the single-threaded NumPy restatement
    nz_cpu = max(2, min(Nz, int(round(3.3e7 / (M * Ny * Nx)))))
    sub = (nz_cpu, M, Ny, Nx)
    rng = np.random.default_rng(0)
    x0 = (100.0 * rng.random(sub)).astype(np.float32)
    n_it = 2
    t0 = time.perf_counter()
    orc.chambolle_pock(x0, n_it, 25.0, scheme=scheme, reg_z_over_reg=reg_z, reg_time=reg_time)
    dt = (time.perf_counter() - t0) / n_it
    vox = float(np.prod(sub))
    mvox_s = vox / dt / 1e6
    full = float(np.prod(shape))
    return {"value": mvox_s * 1e6 / full, "unit": "it/s", "cores": 1, "kind": "port",
            "mvox_per_s": mvox_s,
            "sample": "oracle.chambolle_pock (NumPy, single-threaded like the reference) on a %s z-sub-slab, "
                      "%d iterations, %.1f s/iteration; it/s extrapolated linearly in Nz to %s" % (sub, n_it, dt, tuple(shape)),
            "host_cpus": os.cpu_count(), "numpy": np.__version__}


def cpu_baseline_admm(shape, reg_z, reg_time, scheme, rho, n_cg):
    """oracle.admm (NumPy, single-threaded, Chebyshev x-solve: the recurrence pytv.solvers.ADMM runs by default) timed on a bounded
    sub-volume of the same workload: 2 planes, all M frames, as many whole rows as ~20 s of NumPy allow (the restatement runs at
    ~1 Mvoxel/s for the Nd = 4 schemes and ~0.35 for hybrid); outer iterations / s extrapolated linearly in the voxel count."""
    from oracle import tv_oracle as orc
    Nz, M, Ny, Nx = shape
    target = (7.0e6 if scheme == "hybrid" else 1.8e7)
    nz_cpu = min(Nz, 2)
    rows = int(target / (nz_cpu * M * Nx))
    rows = max(16, min(Ny, rows // 16 * 16))
    sub = (nz_cpu, M, rows, Nx)
    rng = np.random.default_rng(0)
    x0 = (100.0 * rng.random(sub)).astype(np.float32)
    t0 = time.perf_counter()
    orc.admm(x0, 1, 25.0, rho, n_cg, scheme=scheme, reg_z_over_reg=reg_z, reg_time=reg_time, x_solver="chebyshev")
    dt = time.perf_counter() - t0
    vox = float(np.prod(sub))
    return {"value": vox / dt / float(np.prod(shape)), "unit": "outer it/s", "cores": 1, "kind": "port", "mvox_per_s": vox / dt / 1e6,
            "sample": "oracle.admm (NumPy, single-threaded, x_solver='chebyshev', n_cg=%d, rho=%g) on a %s sub-volume (2 planes, all frames, "
                      "%d of %d rows), 1 outer iteration, %.1f s; outer it/s extrapolated linearly in the voxel count to %s"
                      % (n_cg, rho, sub, rows, Ny, dt, tuple(shape)),
            "host_cpus": os.cpu_count(), "numpy": np.__version__}


def cpu_baseline_openmp(shape, reg_z, reg_time):
    """The C / OpenMP oracle (oracle/tv_oracle_c.c, cross-checked against the NumPy oracle) on all host cores."""
    os.environ.setdefault("OMP_PROC_BIND", "spread")          # before libgomp starts its team
    os.environ.setdefault("OMP_PLACES", "threads")
    from oracle import tv_oracle_c as occ
    Nz, M, Ny, Nx = shape
    nz_cpu = max(2, min(Nz, int(round(2.7e8 / (M * Ny * Nx)))))
    sub = (nz_cpu, M, Ny, Nx)
    rng = np.random.default_rng(0)
    x0 = (100.0 * rng.random(sub)).astype(np.float32)
    occ.chambolle_pock(x0[:2], 1, 25.0, scheme="hybrid", reg_z_over_reg=reg_z, reg_time=reg_time)      # build + warm up
    n_it = 3
    # working arrays allocated and first touched by the worker threads (NUMA), time of the iterations alone
    _, _, secs = occ.chambolle_pock(x0, n_it, 25.0, scheme="hybrid", reg_z_over_reg=reg_z, reg_time=reg_time, numa=True)
    dt = secs / n_it
    vox = float(np.prod(sub))
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": vox / dt / float(np.prod(shape)), "unit": "it/s", "cores": threads, "kind": "port", "mvox_per_s": vox / dt / 1e6,
            "sample": "oracle C/OpenMP chambolle_pock (fp32, arrays first touched by the worker threads) on a %s z-sub-slab, %d iterations, "
                      "%.2f s/iteration; it/s extrapolated linearly in Nz to %s" % (sub, n_it, dt, tuple(shape))}


def live_traffic(args):
    """HBM bytes per launch of the CP kernels from rocprofv3 PMC counters, measured NOW: two child runs of this very command
    (--steps 2 --warmup 1, no CPU baseline), one per counter because FETCH_SIZE and WRITE_SIZE do not fit one pass
    (MI355X_MICROARCH.md, HBM / rocprofv3 sections: separate --pmc passes, no tracing next to them; read bytes = 2 x FETCH_SIZE KiB
    on gfx950, write bytes = WRITE_SIZE KiB -- both checked against kernels with exactly known byte counts,
    profiles/r2_pmc_calibration.txt).  Runs AFTER the timed region, once this process has released its device memory.
    Returns ({kernel key: bytes per launch}, note) or (None, reason)."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "already running under a profiler"
    # the child passes run the SAME block structure as the timed region (one block of W, one of K iterations): since round 5 the last
    # iteration of a block differs from the others (its fix-up reads x0: TV_CP_FID_BOTH), a 2 + 1 child would average 2 such fix-ups in 3
    c_steps, c_warm = (min(args.steps, 24), min(args.warmup, 6)) if args.solver == "cp" else (2, 1)
    n_child = c_steps + c_warm                   # launches of every once-per-step kernel in a child pass
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(c_steps), "--warmup", str(c_warm), "--no-cpu-baseline", "--pmc", "off",
             "--workload", args.workload, "--scheme", args.scheme, "--pitch", args.pitch, "--tune-placement", "off",
             "--solver", args.solver, "--rho", repr(args.rho), "--n-cg", str(args.n_cg), "--nz", str(args.nz)] + (["--two-kernel"] if args.two_kernel else []) \
        + (["--allow-single"] if args.allow_single else [])
    sums, totals = {}, {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tvpmc_", dir="/tmp")
        try:
            proc = subprocess.Popen([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child, cwd="/tmp",
                                    env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=240 if counter == "FETCH_SIZE" else 120)      # the first pass may be the box's first `import torch`
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group started above
                proc.wait()
                return None, "rocprofv3 --pmc %s pass timed out" % counter
            if rc != 0:
                return None, "rocprofv3 --pmc %s pass exited with %d" % (counter, rc)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and "tv::k_" in r.get("Kernel_Name", "") and "k_reduce" not in r["Kernel_Name"]:
                        per.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
            if not per:
                return None, "no counter rows for the tv:: kernels in the %s pass" % counter
            for k, v in per.items():
                kib = sum(v) / len(v)
                sums.setdefault(k, {})[counter] = (2.0 if counter == "FETCH_SIZE" else 1.0) * kib * 1024.0
                totals.setdefault(k, {})[counter] = (2.0 if counter == "FETCH_SIZE" else 1.0) * sum(v) * 1024.0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {"fused": 0.0, "fixup": 0.0, "dual": 0.0, "primal": 0.0, "xsolve": 0.0}
    for k, c in sums.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        b = c["FETCH_SIZE"] + c["WRITE_SIZE"]
        if "k_cp_fused" in k:
            out["fused"] += b
        elif "k_cp_fixup" in k:
            out["fixup"] += b
        elif "CpDual" in k:
            out["dual"] += b
        elif "CpPrimal" in k:
            out["primal"] += b
        elif "k_normal_stream" in k or "k_cheb" in k or "k_axpby" in k:
            # the x-solve of an ADMM outer iteration is several launches of several instantiations: bytes per OUTER ITERATION
            out["xsolve"] += (totals[k]["FETCH_SIZE"] + totals[k]["WRITE_SIZE"]) / n_child
    note = ("LIVE: two rocprofv3 child passes of this command (--pmc FETCH_SIZE, --pmc WRITE_SIZE; --steps %d --warmup %d) in %.0f s; "
            "bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section; both counters are exact on "
            "kernels with known byte counts, profiles/r2_pmc_calibration.txt)" % (c_steps, c_warm, time.perf_counter() - t0))
    return {k: (v if v > 0 else None) for k, v in out.items()}, note


def gpu_state(index=0):
    """Clocks / power / temperature of GPU `index` read from sysfs (plain file reads: no HIP call, no child process, nothing that
    talks to the SMU on its own).  Taken right before and right after the timed region so that a run that was clock- or
    power-limited can be told from one that was not (round-3 verdict: the driver's fresh-box runs were 9 % slower than profiles/)."""
    import glob
    out = {}
    # HIP device -> DRM node through the PCI bus id (a lexicographic sort of card0, card1, card10, ... is not the HIP order; round-4 advice);
    # best effort: where the bus id cannot be had, the index into the sorted list is used and the field says so
    cards = []
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        cards = glob.glob("/sys/bus/pci/devices/%s/drm/card[0-9]*/device/pp_dpm_sclk" % bdf)
        if cards:
            out["drm_node"] = "by PCI bus id %s" % bdf
            index = 0
    except Exception:
        cards = []
    if not cards:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"),
                       key=lambda p: int("".join(ch for ch in p.split("/card")[1].split("/")[0] if ch.isdigit()) or 0))
        if cards:
            out["drm_node"] = "best effort: entry %d of the numerically sorted DRM cards" % min(index, len(cards) - 1)
    if not cards:
        # no sysfs view of the GPU (container): one short-lived rocm-smi child instead
        import shutil, subprocess
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        try:
            txt = subprocess.run([exe, "-d", str(index), "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True,
                                 text=True, timeout=20).stdout
            return {"rocm_smi": json.loads(txt)}
        except Exception as exc:
            return {"error": "no /sys/class/drm/card*/device/pp_dpm_sclk and rocm-smi failed: %s" % str(exc)[:120]}
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])

    def rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except Exception:
            return None

    def cur(path):          # "0: 132Mhz\n1: 2400Mhz *" -> the starred level (MHz)
        txt = rd(path)
        if not txt:
            return None
        for line in txt.splitlines():
            if line.rstrip().endswith("*"):
                try:
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                except Exception:
                    return line.strip()
        return txt[:60]

    out["sclk_mhz"] = cur(os.path.join(dev, "pp_dpm_sclk"))
    out["mclk_mhz"] = cur(os.path.join(dev, "pp_dpm_mclk"))
    out["fclk_mhz"] = cur(os.path.join(dev, "pp_dpm_fclk"))
    out["socclk_mhz"] = cur(os.path.join(dev, "pp_dpm_socclk"))
    out["perf_level"] = rd(os.path.join(dev, "power_dpm_force_performance_level"))
    out["gpu_busy_percent"] = rd(os.path.join(dev, "gpu_busy_percent"))
    out["mem_busy_percent"] = rd(os.path.join(dev, "mem_busy_percent"))
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("power1_cap", "power_cap_w", 1e-6),
                                 ("temp1_input", "temp_edge_c", 1e-3), ("temp2_input", "temp_junction_c", 1e-3), ("temp3_input", "temp_mem_c", 1e-3),
                                 ("freq1_input", "freq1_mhz", 1e-6), ("freq2_input", "freq2_mhz", 1e-6)):
            v = rd(os.path.join(hw, name))
            if v is not None and key not in out:
                try:
                    out[key] = round(float(v) * scale, 2)
                except Exception:
                    pass
    return out


def series_summary(v):
    """first 5 / last 5 / min / median / max of a per-iteration series (ms)"""
    v = [float(a) for a in v]
    r3 = lambda a: [round(b, 3) for b in a]
    return {"n": len(v), "first5": r3(v[:5]), "last5": r3(v[-5:]), "min": round(min(v), 3), "median": round(float(np.median(v)), 3),
            "max": round(max(v), 3)}


def _test_stall(rank, phase):
    """TEST knob TV_BENCH_TEST_STALL="<rank>:<phase>": that rank stops making progress at that point (the watchdog test)."""
    spec = os.environ.get("TV_BENCH_TEST_STALL", "")
    if spec == "%d:%s" % (rank, phase):
        sys.stderr.write("[bench] TEST: rank %d stalls before phase '%s'\n" % (rank, phase))
        sys.stderr.flush()
        while True:
            time.sleep(1.0)


def run_admm(args, wl, shape, slab, device, rank, world, local_rank, backend, comm_name, x0, want_live, wd):
    """--solver admm: K outer iterations of pytv.solvers.ADMM (defaults: one-sweep dual side, Chebyshev x-solve, keep_z=True) between
    barriers; HIP events at the phase boundaries of every outer iteration (x-solve | sweep | fix-up); rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    import pytv
    K, W = args.steps, args.warmup
    kw = {}
    if args.pitch == "none":
        kw["pitch"] = None
    if args.tune_placement != "default":
        kw["tune_placement"] = (args.tune_placement == "on")
    ad = pytv.solvers.ADMM(x0, 25.0, args.rho, n_cg=args.n_cg, scheme=args.scheme, reg_z_over_reg=wl["reg_z"], reg_time=wl["reg_time"],
                           slab=slab, **kw)
    nd = ad.geo.nd
    hist = torch.zeros((K + W, 2), dtype=torch.float64, device=device)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(W):
        ad.step(hist[k])
    ad.timing = []
    state_before = gpu_state(local_rank) if rank == 0 else None
    wd.arm("timed region", 120 + 10 * K)
    _test_stall(rank, "timed")
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        ad.step(hist[W + k])
    barrier()
    elapsed = time.perf_counter() - t0
    wd.arm("closing collectives", 600)
    state_after = gpu_state(local_rank) if rank == 0 else None
    marks = ad.timing[:K]
    ad.timing = None
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    hist_r = hist if backend == "nccl" else hist.cpu()
    rccl_ranks = 0
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(hist_r, op=dist.ReduceOp.SUM)
        rccl_ranks = dist.get_world_size() if dist.get_backend() == "nccl" else 0
    elapsed = float(tmax.item())
    h = hist_r.cpu().numpy()
    loss = 0.5 * h[:, 1] + 25.0 * h[:, 0]
    V = float(np.prod(shape))
    V_local = float(slab.nz * shape[1] * shape[2] * shape[3])
    it_s = K / elapsed
    torch.cuda.synchronize()
    # phase durations from the marks: [start .. xsolve] = x-solve, [xsolve .. sweep] = tv_admm_fused, [sweep .. end] = halo + tv_admm_fixup
    def span(a, b):
        v = []
        for m in marks:
            d = dict(m)
            if a in d and b in d:
                v.append(d[a].elapsed_time(d[b]))
        return v
    s_x, s_sw, s_fx = span("start", "xsolve"), span("xsolve", "sweep"), span("sweep", "end")
    fused_path = bool(ad.fused and ad.cheb and s_sw)
    keep_z, xs_desc, xs_launches = bool(ad.keep_z), ad.xsolve_desc, ad.xsolve_launches
    words_sweep = 2 * nd + 3        # u read + written, x, x0 read, r written (since round 5 also with keep_z: z is rebuilt on demand from a second u array)
    words_x = ad.xsolve_words
    out = {
        "metric": METRIC["admm"], "value": it_s, "unit": "it/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %s fp32 %s ADMM" % (args.workload, "x".join(str(s) for s in shape), args.scheme),
                   "shape": list(shape), "scheme": args.scheme, "nd": nd, "reg_z_over_reg": wl["reg_z"], "reg_time": wl["reg_time"],
                   "lambda": 25.0, "rho": args.rho, "n_cg": args.n_cg, "x_solver": "chebyshev" if ad.cheb else "cg", "keep_z": keep_z,
                   "kernels": ("one-sweep dual side: tv_admm_fused + tv_admm_fixup; x-solve: %s" % xs_desc) if fused_path
                              else "unfused: tv_admm_tu + tv_DT_axpy + tv_normal_op2 / tv_cheb_step",
                   "parallelism": "z-slab x%d" % world},
        "voxel_iterations_per_sec": it_s * V,
        "words_per_voxel_and_outer_iteration": {"sweep": words_sweep, "xsolve": words_x, "total": words_sweep + words_x},
        "hbm_gbps_iteration": {"algorithmic": 4.0 * (words_sweep + words_x) * V * it_s / 1e9 / world, "per": "GPU"},
        "loss_first_last": [float(loss[W]), float(loss[-1])],
        "placement_tuning": getattr(ad, "placement", None),
        "rccl_ranks": rccl_ranks, "comm": comm_name,
        "halo": {"backend": (dist.get_backend() if dist.is_initialized() else None), "bytes_per_plane": 4 * shape[1] * shape[2] * shape[3]},
        "gpu_state": {"before": state_before, "after": state_after, "source": "sysfs (pp_dpm_*, hwmon) read by rank 0 outside the timed region"},
    }
    if fused_path:
        out["series_ms"] = {"xsolve": series_summary(s_x), "sweep": series_summary(s_sw), "fixup": series_summary(s_fx),
                            "note": "HIP events on the launch stream at the phase boundaries of every outer iteration"}
    live, live_note = None, "--pmc off"
    wd.arm("post-run: live PMC passes, CPU baseline on rank 0, final barrier", 1800)
    if want_live:
        del ad, x0, hist, hist_r, marks
        torch.cuda.empty_cache()
        try:
            live, live_note = live_traffic(args)
        except Exception as e:
            live, live_note = None, "live PMC passes failed: %r" % (e,)
    traffic = live or {}
    src = live_note if live is not None else "not measured in this run (%s)" % live_note
    sharded = " (per GPU)" if world > 1 else ""
    if fused_path:
        t_sw, t_x, t_fx = float(np.mean(s_sw)) * 1e-3, float(np.mean(s_x)) * 1e-3, float(np.mean(s_fx)) * 1e-3
        b_sw, b_x = 4.0 * words_sweep * V_local, 4.0 * words_x * V_local
        # the dominant kernel of the outer iteration: the one sweep over u (z / u update + the residual of the next x-solve)
        out["roofline"] = {"bound": "hbm", "kernel": "tv_admm_fused: k_cp_fused<S,M,...,ALG_ADMM> (group soft threshold + u update + lagged r = (x0 - x) + rho D^T t')",
                           "achieved": b_sw / t_sw / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_sw / t_sw / 1e9 / HBM_PEAK_GBPS,
                           "traffic": traffic.get("fused"), "traffic_source": src, "bytes_per_launch": b_sw, "ms_per_launch": 1e3 * t_sw,
                           "note": "algorithmic bytes (2 Nd + 3) * 4 per voxel: u read + written%s, x, x0 read, r written; HIP events on the launch stream"
                                   % (" (ping-pong: keep_z rebuilds z on demand)" if keep_z else "") + sharded}
        out["roofline_xsolve"] = {"bound": "hbm", "kernel": xs_desc, "achieved": b_x / t_x / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": b_x / t_x / 1e9 / HBM_PEAK_GBPS, "traffic": traffic.get("xsolve"), "bytes_per_outer_iteration": b_x,
                                  "ms_per_outer_iteration": 1e3 * t_x, "launches_per_outer_iteration": xs_launches,
                                  "note": "all launches of the x-solve of ONE outer iteration together: %d words per voxel%s" % (words_x, sharded)}
        out["roofline_fixup"] = {"kernel": "halo exchange of t' + tv_admm_fixup", "ms_per_launch": 1e3 * t_fx, "traffic": traffic.get("fixup")}
    if dist.is_initialized():
        dist.barrier()
    if rank == 0 and not args.no_cpu_baseline:
        if not want_live:
            del ad, x0
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_admm(shape, wl["reg_z"], wl["reg_time"], args.scheme, args.rho, args.n_cg)
    if rank == 0:
        print(json.dumps(out), flush=True)
    wd.arm("process group teardown", 120)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    wd.disarm()


def error_line(args, n_gpus, message, **extra):
    """The JSON line of a run that produced no number: same leading keys as a good line, value null, the reason in `error`."""
    out = {"metric": METRIC[args.solver], "value": None, "unit": "it/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": args.workload}}
    out.update(extra)
    out["error"] = message
    return json.dumps(out)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) from a plain shell, no RANK / WORLD_SIZE in the environment: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>` as a CHILD process, relay its
    stdout (rank 0's JSON line) and return its exit code.  This process never imports torch and never makes a HIP call -- it is not
    replaced either (no exec).  A child that outlives TV_BENCH_TIMEOUT seconds is killed by process group and reported as an error line;
    a child that fails without a line gets one.  A failed attempt is repeated with a more conservative transport (TV_BENCH_RETRY=0
    switches that off): first without the halo / compute overlap (--no-overlap), then with host-staged halos over gloo, every rank
    still on its own GPU (TV_BENCH_BACKEND=gloo: a number that is labelled as such in `comm`, instead of none).  stderr says which
    attempt produced the line."""
    import signal
    import socket
    import subprocess
    import threading
    assert "torch" not in sys.modules, "the launcher must start before torch is imported"
    n = args.gpus
    module = os.environ.get("TV_BENCH_LAUNCH_MODULE", "torch.distributed.run")        # (tests substitute a recorder)
    limit = float(os.environ.get("TV_BENCH_TIMEOUT", "2400"))
    attempts = [(list(argv), {})]
    if os.environ.get("TV_BENCH_RETRY", "1") != "0":
        if "--no-overlap" not in argv:
            attempts.append((list(argv) + ["--no-overlap"], {}))
        if os.environ.get("TV_BENCH_BACKEND", "nccl") == "nccl":
            attempts.append(([a for a in argv if a != "--no-overlap"] + ["--no-overlap"], {"TV_BENCH_BACKEND": "gloo"}))
    rc = 1
    for k, (av, env_extra) in enumerate(attempts):
        port = os.environ.get("TV_BENCH_MASTER_PORT")
        if port is None:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", module, "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.abspath(__file__)] + av
        env = dict(os.environ, TV_BENCH_SELF_LAUNCHED="1", **env_extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        sys.stderr.write("[bench] self-launch (torch imported: %s): %s\n" % ("torch" in sys.modules, " ".join(cmd)))
        sys.stderr.flush()
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
        got_line = [False]

        def relay():
            for ln in proc.stdout:
                if ln.startswith("{"):
                    got_line[0] = True
                sys.stdout.write(ln)
                sys.stdout.flush()

        th = threading.Thread(target=relay, daemon=True)
        th.start()
        try:
            rc = proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group started above
            except ProcessLookupError:
                pass
            proc.wait()
            th.join(10)
            print(error_line(args, n, "the %d-rank job did not finish within TV_BENCH_TIMEOUT = %.0f s and was killed" % (n, limit)), flush=True)
            return 5
        th.join(10)
        if rc == 0 and got_line[0]:
            return 0
        if not got_line[0]:
            print(error_line(args, n, "the launcher (%s) exited with code %d and rank 0 printed no line" % (module, rc)), flush=True)
        if k + 1 < len(attempts):
            nxt = attempts[k + 1]
            sys.stderr.write("[bench] attempt %d failed (exit code %d); next attempt: %s %s\n" % (k + 1, rc, " ".join(nxt[0]), nxt[1] or ""))
    return rc if rc != 0 else 1


class Watchdog:
    """A stuck collective / `work.wait()` / kernel must not hang the job: every rank arms a deadline per PHASE of the run
    (communicator setup, construction + warm-up, the timed region, the per-phase pass, the closing collectives).  A rank whose deadline
    passes writes the phase and the deadline to stderr, rank 0 prints an error JSON line, and the process ends with os._exit(4) -- the
    launcher then takes the other ranks down.  Nothing is re-exec'ed and no GPU call is made from the watchdog thread.
    TV_BENCH_WATCHDOG_SCALE multiplies every deadline (0 disables)."""

    def __init__(self, args, rank, world):
        import threading
        self.args, self.rank, self.world = args, rank, world
        self.scale = float(os.environ.get("TV_BENCH_WATCHDOG_SCALE", "1"))
        self.lock = threading.Lock()
        self.phase, self.deadline = None, None
        if self.scale > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def arm(self, phase, seconds):
        with self.lock:
            self.phase, self.deadline = phase, time.monotonic() + seconds * self.scale

    def disarm(self):
        with self.lock:
            self.phase, self.deadline = None, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                phase, deadline = self.phase, self.deadline
            if deadline is not None and time.monotonic() > deadline:
                msg = "watchdog: rank %d of %d made no progress in phase '%s' within its deadline" % (self.rank, self.world, phase)
                sys.stderr.write("[bench] %s\n" % msg)
                sys.stderr.flush()
                if self.rank == 0:
                    print(error_line(self.args, self.world, msg), flush=True)
                os._exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="northstar", choices=sorted(WORKLOADS))
    ap.add_argument("--scheme", default="hybrid", choices=["upwind", "downwind", "central", "hybrid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--two-kernel", action="store_true", help="force the dual + primal kernel pair instead of the one-sweep kernel")
    ap.add_argument("--comm", default="torch", choices=["torch", "cabi"],
                    help="halo exchange through torch.distributed (RCCL process group, default) or through the C-ABI's own RCCL "
                         "context (tv_ctx_create / tv_halo_exchange; torch.distributed only hands out the unique id)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto (1 GPU only): measure roofline.traffic live with two rocprofv3 --pmc child passes after the timed run; "
                         "off: copy the committed numbers of profiles/traffic.json and label them STATIC")
    ap.add_argument("--prewarm-gb", type=float, default=float(os.environ.get("TV_BENCH_PREWARM_GB", "0")),
                    help="touch and release this many GB of device memory before the volume is allocated (placement experiment)")
    ap.add_argument("--pitch", default=os.environ.get("TV_BENCH_PITCH", "default"),
                    help="layout of the solver's private state: default (what solvers.ChambollePock picks), none (dense), auto, or "
                         "<frame pad in bytes> (rows rounded up to 128 B, frames padded by that many bytes)")
    ap.add_argument("--tune-placement", default="default", choices=["default", "on", "off"],
                    help="solvers.ChambollePock(tune_placement=...): default = the solver's own rule")
    ap.add_argument("--phases", action="store_true", help="per-phase HIP-event times of the schedule in the JSON line (always on at N > 1)")
    ap.add_argument("--solver", default=None, choices=["cp", "admm"], help="default: what the workload names (config4* / admm-small: admm)")
    ap.add_argument("--rho", type=float, default=0.05, help="ADMM penalty parameter")
    ap.add_argument("--n-cg", type=int, default=5, help="ADMM: steps of the x-solve per outer iteration")
    ap.add_argument("--nz", type=int, default=0,
                    help="TEST knob: run the workload with this many planes instead of its own (the rehearsals of the 8-GPU configs on one "
                         "GPU, tests/test_gpu_rccl.py); the line's config.shape and config.workload say so")
    ap.add_argument("--allow-single", action="store_true",
                    help="run a workload meant for several GPUs (config3, config4) on fewer ranks than it names, memory permitting")
    args = ap.parse_args()
    if args.solver is None:
        args.solver = WORKLOADS[args.workload].get("solver", "cp")
    # N > 1 from a plain shell: this process becomes the launcher's parent BEFORE torch is imported or any HIP call is made
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    # multi-GPU workloads are refused on fewer ranks BEFORE anything touches the GPU (exit code 2, one JSON line with the reason)
    _wl, _world = WORKLOADS[args.workload], int(os.environ.get("WORLD_SIZE", "1"))
    if _world < _wl.get("min_gpus", 1) and not args.allow_single:
        if int(os.environ.get("RANK", "0")) == 0:
            print(error_line(args, _world, "workload %s %s is a multi-GPU job (>= %d ranks); pass --allow-single to run it on %d"
                             % (args.workload, "x".join(map(str, _wl["shape"])), _wl["min_gpus"], _world)), flush=True)
        sys.exit(2)
    # must be in the environment BEFORE the HIP runtime starts (the pool's driver only supports dmabuf IPC; RCCL's
    # cross-process buffer sharing fails with hipIpcGetMemHandle: invalid argument otherwise).  Round 2 set it after
    # torch.cuda.set_device(), where it could no longer take effect.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    live, live_note = None, "--pmc off"
    want_live = False
    if args.pmc == "auto":
        want_live = (args.gpus == 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1
                     and os.environ.get("TV_BENCH_PMC", "1") != "0")
        if not want_live:
            live_note = "PMC passes run with one GPU only"

    import torch
    import torch.distributed as dist
    import pytv
    from pytv.slab import Slab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(args, rank, world)
    wd.arm("communicator setup", 420)
    # TEST-ONLY knobs (tests / single-GPU box): several ranks on one GPU over gloo with host-staged halos
    backend = os.environ.get("TV_BENCH_BACKEND", "nccl")
    if os.environ.get("TV_BENCH_SHARE_GPU", "0") == "1":
        local_rank = 0
    local_rank %= max(1, torch.cuda.device_count())      # launchers that pin one visible device per rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    wl = WORKLOADS[args.workload]
    shape = wl["shape"]
    if args.nz > 0:
        shape = (args.nz,) + tuple(shape[1:])
    if args.prewarm_gb > 0:
        junk = torch.empty(int(args.prewarm_gb * (1 << 30)), dtype=torch.uint8, device=device)
        junk.fill_(1)
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
    native = None
    comm_name = "none (one process, no process group)"
    try:
        if world > 1 or "RANK" in os.environ:
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            tmo = datetime.timedelta(seconds=int(os.environ.get("TV_BENCH_INIT_TIMEOUT", "300")))
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)   # RCCL on ROCm
                comm_name = "torch.distributed nccl (= RCCL): batch_isend_irecv halos, device all-reduces"
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
                comm_name = "torch.distributed %s (TEST backend: host-staged halos)" % backend
        if args.comm == "cabi" and dist.is_initialized() and world > 1:
            from pytv.slab import NativeComm
            native = NativeComm(device=device)
            comm_name = "C-ABI RCCL context (tv_ctx_create / tv_halo_exchange / tv_allreduce_f64); unique id handed out over torch.distributed " + dist.get_backend()
        slab = Slab(shape[0], rank=rank, world=world, native_comm=native)
        if dist.is_initialized():
            # first contact with the transport BEFORE any work is queued: a failure here is reported as such
            probe = torch.ones(1, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            assert int(probe.item()) == world, "all-reduce over %d ranks returned %r" % (world, probe.item())
    except Exception as exc:              # noqa: BLE001 -- whatever the transport throws: say so in the JSON line, fail the run
        if rank == 0:
            print(error_line(args, world, "communicator setup failed on rank %d of %d (backend %s, comm %s): %s: %s"
                             % (rank, world, backend, args.comm, type(exc).__name__, str(exc)[:600]), comm=comm_name), flush=True)
        sys.stderr.write("[bench rank %d] communicator setup failed: %r\n" % (rank, exc))
        sys.stderr.flush()
        os._exit(3)                       # every rank that fails exits non-zero at once (no destructor tries the dead communicator)
    wd.arm("input synthesis, solver construction, warm-up", 900)
    x0 = synth_slab(shape, slab.z0, slab.nz, device)
    if args.solver == "admm":
        run_admm(args, wl, shape, slab, device, rank, world, local_rank, backend, comm_name, x0, want_live, wd)
        return
    pkw = {}
    if args.pitch == "none":
        pkw["pitch"] = None
    elif args.pitch == "auto":
        pkw["pitch"] = "auto"
    elif args.pitch != "default":
        pkw["pitch"] = pytv.solvers.auto_pitch(shape[2], shape[3], torch.float32, frame_pad_bytes=int(args.pitch))
    if args.tune_placement != "default":
        pkw["tune_placement"] = (args.tune_placement == "on")
    cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=args.scheme, reg_z_over_reg=wl["reg_z"], reg_time=wl["reg_time"],
                                    slab=slab, overlap=not args.no_overlap, fused=False if args.two_kernel else None, **pkw)
    state_layout = {"row_pitch_elems": cp.geo.row_pitch, "frame_pitch_elems": cp.geo.frame_pitch,
                    "frame_pad_bytes": 4 * (cp.geo.frame_pitch - shape[2] * cp.geo.row_pitch), "pitched": bool(cp.geo.pitched)}
    nd = cp.geo.nd
    K, W = args.steps, args.warmup
    hist = torch.zeros((K + W, cp.SLOTS), dtype=torch.float64, device=device)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    cp.run_steps(hist[:W])
    # per-kernel HIP events on the launch stream (torch's current stream == the stream the C-ABI enqueues on)
    cp.timing = []
    want_phases = args.phases or world > 1

    # (sampled BEFORE the barrier: a sysfs / rocm-smi read on rank 0 between the barrier and t0 would be time the other ranks spend
    # waiting in their first halo exchange -- round-4 advice)
    state_before = gpu_state(local_rank) if rank == 0 else None
    wd.arm("timed region", 120 + 10 * K)
    _test_stall(rank, "timed")
    barrier()
    t0 = time.perf_counter()
    cp.run_steps(hist[W:W + K])         # K iterations; the last sweep returns the final iterate's fidelity too (TV_CP_FID_BOTH): no extra pass
    barrier()
    elapsed = time.perf_counter() - t0
    wd.arm("per-phase pass and closing collectives", 600)
    state_after = gpu_state(local_rank) if rank == 0 else None
    ev = cp.timing[:K]
    cp.timing = None
    phases = None
    if want_phases:
        # per-phase events in a SEPARATE short pass after the timed region: the headline ms_per_step carries the same (three events
        # per step) instrumentation at every N, so the 1 -> N ratio is not biased by ~8 extra event records per step (round-3 advice)
        cp.phase_timing = []      # one event per phase boundary of the schedule (interior / wait / edges ...), see solvers.py
        scratch = torch.zeros((cp.SLOTS,), dtype=torch.float64, device=device)
        n_ph = max(2, min(K, 6))
        n_ph += n_ph & 1                                  # even: the x ping-pong ends where the timed loop left it
        for it in range(n_ph):
            cp.step(scratch)
        barrier()
        pm = pytv.solvers.ChambollePock.phase_means_ms(cp.phase_timing[:n_ph])
        cp.phase_timing = None
        names = list(pm)
        pt = torch.tensor([pm[k] for k in names], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        pmax, psum = pt.clone(), pt.clone()
        if dist.is_initialized():
            dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(psum, op=dist.ReduceOp.SUM)
        phases = {"unit": "ms per step", "max_over_ranks": {k: float(v) for k, v in zip(names, pmax.tolist())},
                  "mean_over_ranks": {k: float(v) / world for k, v in zip(names, psum.tolist())},
                  "rank0": pm,
                  "steps": n_ph,
                  "note": "collected in a separate pass of `steps` iterations AFTER the timed region (the timed loop carries no phase events); "
                          "HIP events on the launch stream at every phase boundary of the schedule (pytv/solvers.py _step_fused / step); "
                          "*_halo_wait_exposed = time the launch stream sat in wait(handle) with nothing left to overlap (0 when the "
                          "transfer finished behind the interior work); the per-rank sums need not equal ms_per_step (max over ranks of the whole run)"}

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    hist_r = hist if backend == "nccl" else hist.cpu()
    rccl_ranks = 0
    if dist.is_initialized():           # also with ONE rank: the device all-reduce then runs on RCCL like it does with 8
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(hist_r, op=dist.ReduceOp.SUM)
        rccl_ranks = dist.get_world_size() if dist.get_backend() == "nccl" else 0
    elapsed = float(tmax.item())
    h = hist_r.cpu().numpy()
    loss = cp.loss_from_slots(h, 25.0)

    V = float(np.prod(shape))
    V_local = float(slab.nz * shape[1] * shape[2] * shape[3])
    it_s = K / elapsed
    bytes_iter_algo = 4.0 * (8 + 3 * nd) * V            # the README's un-fused iteration
    out = {
        "metric": METRIC["cp"], "value": it_s, "unit": "it/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %s fp32 %s CP" % (args.workload, "x".join(str(s) for s in shape), args.scheme),
                   "shape": list(shape), "scheme": args.scheme, "nd": nd, "reg_z_over_reg": wl["reg_z"], "reg_time": wl["reg_time"],
                   "lambda": 25.0, "sigma_D": cp.sigma_D, "sigma_A": cp.sigma_A, "tau": cp.tau,
                   "parallelism": "z-slab x%d%s" % (world, " (halo overlapped)" if (cp.overlap or getattr(cp, "overlap_fused", False)) else "")},
        "voxel_iterations_per_sec": it_s * V,
        # bytes the kernels of THIS run must move per iteration (one sweep: 5 + 2 Nd words; kernel pair: 6 + 3 Nd) over the wall time
        "hbm_gbps_iteration": {"algorithmic": 4.0 * ((5 + 2 * nd) if cp.fused else (6 + 3 * nd)) * V * it_s / 1e9 / world, "per": "GPU",
                               "words_per_voxel": (5 + 2 * nd) if cp.fused else (6 + 3 * nd)},
        # NOT a bandwidth: the rate the reference's un-fused iteration ((8 + 3 Nd) words) would need to keep this pace
        "equivalent_unfused_gbps": {"readme_(8+3Nd)_words": bytes_iter_algo * it_s / 1e9 / world, "per": "GPU",
                                    "note": "effective rate of work the fused kernels do not do; may exceed the HBM peak"},
        "loss_first_last": [float(loss[W]), float(loss[-1])],
        "state_layout": state_layout,
        # solvers.ChambollePock picks the two buffers of the x ping-pong by measurement at construction (outside the timed region, like
        # the allocation itself): the candidates' sweep times and the pair it kept
        "placement_tuning": getattr(cp, "placement", None),
        "rccl_ranks": rccl_ranks, "comm": comm_name,        # rccl_ranks: size of the RCCL communicator the run used (0: no process group / test backend)
        "halo": {"backend": (dist.get_backend() if dist.is_initialized() else None), "planes_per_exchange": 1,
                 "bytes_per_plane": 4 * shape[1] * shape[2] * shape[3], "exchanges_per_iteration": 2 if world > 1 else 0},
    }
    if phases is not None:
        out["phases"] = phases
    traffic = {}
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("%s|%s" % (args.workload, args.scheme), {})
    except Exception:
        pass
    torch.cuda.synchronize()
    s_k1 = [e[0].elapsed_time(e[1]) for e in ev]
    s_k2 = [e[1].elapsed_time(e[2]) for e in ev]
    t_k1 = float(np.mean(s_k1)) * 1e-3
    t_k2 = float(np.mean(s_k2)) * 1e-3
    # per-iteration series of the timed region (HIP events on the launch stream) and the GPU's clocks / power right before and right
    # after it: a drift (clock ramp, throttling) shows in first5 vs last5, a constant offset (placement) does not
    out["series_ms"] = {"kernel1": series_summary(s_k1), "kernel2": series_summary(s_k2),
                        "step": series_summary([a.elapsed_time(b) for (a, _, _), (b, _, _) in zip(ev[:-1], ev[1:])]) if len(ev) > 1 else None,
                        "note": "kernel1 = sweep (or dual), kernel2 = fix-up (or primal), step = start of iteration k to start of k+1"}
    out["gpu_state"] = {"before": state_before, "after": state_after, "source": "sysfs (pp_dpm_*, hwmon) read by rank 0 outside the timed region"}
    cp_fused = bool(cp.fused)
    wd.arm("post-run: live PMC passes, CPU baseline on rank 0, final barrier", 1800)
    if want_live:
        # AFTER the timed region, with this process's device memory handed back: the child passes need the HBM for the same
        # volume, and a process that starts right after another one released ~100 GB can run 5 - 7 % slower for its whole life
        #  -- the children do not care, the timed run above would
        del cp, x0, ev, hist, hist_r
        torch.cuda.empty_cache()
        try:
            live, live_note = live_traffic(args)
        except Exception as e:              # the profiler is evidence, not the product: never let it take the bench down
            live, live_note = None, "live PMC passes failed: %r" % (e,)
    traffic_source = "STATIC, not measured in this run (%s): %s" % (live_note, traffic.get("source"))
    if live is not None:
        traffic, traffic_source = live, live_note
    sharded = " (per GPU; kernel 2 interval includes the halo wait)" if world > 1 else ""
    if cp_fused:
        b_k1 = 4.0 * (5 + 2 * nd) * V_local      # read x, x0, p, q ; write q, x, p
        out["config"]["kernels"] = "one-sweep: tv_cp_fused + tv_cp_fixup"
        out["roofline"] = {"bound": "hbm", "kernel": "tv_cp_fused: k_cp_fused<S,M> (dual update + lagged primal update, one pass over q)",
                           "achieved": b_k1 / t_k1 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_k1 / t_k1 / 1e9 / HBM_PEAK_GBPS,
                           "traffic": traffic.get("fused"),
                           "traffic_source": traffic_source, "bytes_per_launch": b_k1,
                           "ms_per_launch": 1e3 * t_k1,
                           "note": "algorithmic bytes (5+2Nd)*4 per voxel: q read+written once, x/x0/p read, x/p written; HIP events on the "
                                   "launch stream (includes the tiny partial-sum kernels)" + sharded}
        out["roofline_fixup"] = {"kernel": "tv_cp_fixup: k_cp_fixup<S> (tile-edge rows/cols, chunk-edge planes)", "ms_per_launch": 1e3 * t_k2,
                                 "traffic": traffic.get("fixup"),
                                 "note": "adds the adjoint terms that cross wave-tile rows, block-tile columns, z-chunks and slabs (2 of every 8 rows since round 2: ~1.65 words/voxel moved; round 1: 2 of 4, ~2.5)"}
    else:
        b_dual = 4.0 * (1 + 2 * nd) * V_local
        b_primal = 4.0 * (nd + 5) * V_local
        out["config"]["kernels"] = "two-kernel: tv_cp_dual + tv_cp_primal"
        out["roofline"] = {"bound": "hbm", "kernel": "tv_cp_dual: k_D_march<S,M,CpDual> (fp32, planes >= 4 MiB) or k_D<S,T,V,CpDual>",
                           "achieved": b_dual / t_k1 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": b_dual / t_k1 / 1e9 / HBM_PEAK_GBPS,
                           "traffic": traffic.get("dual"),
                           "traffic_source": traffic_source, "bytes_per_launch": b_dual,
                           "ms_per_launch": 1e3 * t_k1,
                           "note": "algorithmic bytes (1+2Nd)*4 per voxel; HIP events on the launch stream (includes the tiny partial-sum kernels)" + sharded}
        out["roofline_primal"] = {"bound": "hbm", "kernel": "tv_cp_primal: k_DT_march<S,M,CpPrimal> or k_DT<S,T,V,SrcPlain,CpPrimal>",
                                  "achieved": b_primal / t_k2 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": b_primal / t_k2 / 1e9 / HBM_PEAK_GBPS, "traffic": traffic.get("primal"),
                                  "bytes_per_launch": b_primal, "ms_per_launch": 1e3 * t_k2,
                                  "note": "algorithmic bytes (Nd+5)*4 per voxel: reads q,x,x0,p; writes x,p (fidelity dual fused in)" + sharded}
    if dist.is_initialized():
        dist.barrier()          # the other ranks wait at the final barrier while rank 0 times the host baseline
    if rank == 0 and not args.no_cpu_baseline:
        if not want_live:
            del cp, x0
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(shape, wl["reg_z"], wl["reg_time"], nd, args.scheme)
        try:
            if world == 1:      # all host cores: only when no other rank's threads share them
                out["cpu_baseline_openmp"] = cpu_baseline_openmp(shape, wl["reg_z"], wl["reg_time"])
        except Exception as exc:                      # no gcc / OpenMP on this host: the NumPy baseline stands alone
            out["cpu_baseline_openmp"] = {"error": str(exc)[:200]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    wd.arm("process group teardown", 120)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    wd.disarm()


if __name__ == "__main__":
    main()

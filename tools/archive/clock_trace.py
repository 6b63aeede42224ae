#!/usr/bin/env python3
"""Round 4, verdict item 1(a): clocks and power DURING the north-star iteration (bench.py reads them only before / after its timed
region, where the GPU has already dropped to idle).  A thread samples sysfs every ~25 ms while the main thread runs the one-sweep
CP loop for a few seconds; printed: the time series (sclk / mclk / fclk / socclk / power / temperatures), the per-iteration sweep
times, and the gpu_metrics blob of the first and the last sample (hex) for offline decoding (throttle status, average clocks).
A diagnostic tool: bench.py's timed region carries no sampler.
usage: python tools/clock_trace.py [seconds=3]"""
import glob, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
t0_import = time.perf_counter()
import numpy as np, torch, pytv
from bench import synth_slab
import_s = time.perf_counter() - t0_import
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0

cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
dev_dir = os.path.dirname(cards[0]) if cards else None
hw = (glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*")) or [None])[0] if dev_dir else None


def rd(path, binary=False):
    try:
        with open(path, "rb" if binary else "r") as f:
            return f.read() if binary else f.read().strip()
    except Exception:
        return None


def cur(path):
    txt = rd(path)
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            return "".join(ch for ch in line.split(":")[1] if ch.isdigit())
    return None


def sample():
    s = {"t": time.perf_counter()}
    if dev_dir:
        for k in ("sclk", "mclk", "fclk", "socclk"):
            s[k] = cur(os.path.join(dev_dir, "pp_dpm_" + k))
        s["busy"] = rd(os.path.join(dev_dir, "gpu_busy_percent"))
    if hw:
        for name, key in (("power1_average", "p_avg"), ("power1_input", "p_in"), ("temp2_input", "t_junc"), ("temp3_input", "t_mem"), ("freq1_input", "f1"), ("freq2_input", "f2")):
            v = rd(os.path.join(hw, name))
            if v is not None:
                s[key] = v
    return s


samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        samples.append(sample())
        time.sleep(0.025)


dev = torch.device("cuda", 0)
x0 = synth_slab((256, 8, 1024, 1024), 0, 256, dev)
cp = pytv.solvers.ChambollePock(x0, 25.0, reg_time=1.0, fused=True)
for _ in range(3):
    cp.step()
torch.cuda.synchronize()
blob0 = rd(os.path.join(dev_dir, "gpu_metrics"), binary=True) if dev_dir else None
cp.timing = []
th = threading.Thread(target=sampler, daemon=True)
t_start = time.perf_counter()
th.start()
n = 0
while time.perf_counter() - t_start < secs:
    for _ in range(4):
        cp.step()
        n += 1
    torch.cuda.synchronize()
blob1 = rd(os.path.join(dev_dir, "gpu_metrics"), binary=True) if dev_dir else None
stop.set()
th.join()
k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]
print("# import %.1f s; %d iterations in %.2f s; sweep ms: mean %.3f even %.3f odd %.3f min %.3f max %.3f" % (
    import_s, n, time.perf_counter() - t_start, np.mean(k1), np.mean(k1[0::2]), np.mean(k1[1::2]), min(k1), max(k1)))
print("# sweep series:", " ".join("%.2f" % v for v in k1))
keys = [k for k in ("sclk", "mclk", "fclk", "socclk", "busy", "p_avg", "p_in", "t_junc", "t_mem", "f1", "f2") if any(k in s for s in samples)]
print("# samples (%d): t_ms " % len(samples) + " ".join(keys))
for s in samples:
    print("%8.1f " % ((s["t"] - t_start) * 1e3) + " ".join(str(s.get(k)) for k in keys))
for name, b in (("first", blob0), ("last", blob1)):
    if b:
        print("# gpu_metrics %s (%d bytes): %s" % (name, len(b), b.hex()))

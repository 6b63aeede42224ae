#!/usr/bin/env python3
"""Round 5, verdict item 5c: does carving x, x_alt, p, x0 and q out of ONE allocation make the placement mode of the one-sweep
Chambolle-Pock kernel deterministic?  profiles/r4_deltatest.txt: inside one allocation the rate of a copy depends reproducibly on the
DISTANCE between its streams (fast at multiples of 4 GiB + 0 and + 4 .. 32 MiB, slow at + 0.5 .. 1 MiB, + 5 MiB, + 64 .. 128 MiB).
Here: N constructions per layout in ONE process; each runs 2 + 6 iterations of the north-star problem (tuner off) and reports the mean
sweep time of the two ping-pong directions.  Layouts: "separate" = five torch allocations (the solver's default), "slab+<gap MiB>" =
one torch.empty for everything, arrays back to back with <gap> MiB between them.
usage: python tools/archive/slab_placement_probe.py [NzxMxNyxNx] [constructions] [layout,layout,...]   layout = separate | slab+<gap MiB>[q|m|a]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab

shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (256, 8, 1024, 1024)
n_con = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
x0_src = synth_slab(shape, 0, shape[0], dev)
V = x0_src.numel()


def run(cp):
    hist = torch.zeros((8, cp.SLOTS), dtype=torch.float64, device=dev)
    cp.run_steps(hist[:2])
    cp.timing = []
    cp.run_steps(hist[2:8])
    torch.cuda.synchronize()
    sw = [e[0].elapsed_time(e[1]) for e in cp.timing]
    cp.timing = None
    return float(np.mean(sw[0::2])), float(np.mean(sw[1::2]))


ORDERS = {"": ("x", "x_alt", "p", "x0", "q"), "q": ("q", "x", "x_alt", "p", "x0"), "m": ("x", "x_alt", "q", "p", "x0"), "a": ("x", "p", "x_alt", "x0", "q")}


PAD_GIB = float(os.environ.get("SLAB_PAD_GIB", "0"))      # allocate (and keep, during the construction) this much BEFORE the slab: shifts where the slab lands


def carve(cp, gap_mib, order=""):
    """rebind the solver's arrays to views of one allocation; order: "" images then q, "q" q first, "m" q in the middle, "a" x p x_alt x0 q"""
    nd = cp.geo.nd
    img, grad = V, V * nd
    gap = gap_mib * (1 << 20) // 4
    total = 5 * gap + 4 * img + grad
    slab = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    views = {}
    for name in ORDERS[order]:
        n, shp = (grad, cp.q.shape) if name == "q" else (img, cp.x.shape)
        views[name] = slab[off:off + n].view(shp)
        off += n + gap
    views["x0"].copy_(x0_src)
    views["x"].copy_(x0_src)
    views["x_alt"].zero_(); views["p"].zero_(); views["q"].zero_()
    cp.x, cp.x_alt, cp.p, cp.x0, cp.q = views["x"], views["x_alt"], views["p"], views["x0"], views["q"]
    return slab


print("shape %s; sweep ms of the two ping-pong directions (mean of 3 each), %d constructions per layout, one process" % (shape, n_con))
LAYOUTS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["separate", "slab+0", "slab+8", "slab+16", "separate", "slab+0", "slab+1"]
for layout in LAYOUTS:
    res = []
    for k in range(n_con):
        cp = pytv.solvers.ChambollePock(x0_src, 25.0, scheme="hybrid", reg_time=1.0, tune_placement=False)
        keep = None
        if layout != "separate":
            tag = layout.split("+")[1]
            pad_gib = 0.0
            if "@" in tag:                       # "slab+32@7": 7 GiB allocated first, the slab after it, the pad freed before the sweeps run
                tag, pg = tag.split("@")
                pad_gib = float(pg)
            pad = torch.empty(int(pad_gib * (1 << 30)), dtype=torch.uint8, device=dev) if pad_gib > 0 else None
            keep = carve(cp, int("".join(ch for ch in tag if ch.isdigit())), "".join(ch for ch in tag if ch.isalpha()))
            del pad
        res.append(run(cp))
        del cp, keep
        torch.cuda.empty_cache()
    flat = [v for r in res for v in r]
    print("%-9s %s   min %.2f max %.2f spread %.1f %%" % (layout, "  ".join("%.2f/%.2f" % r for r in res), min(flat), max(flat), 100 * (max(flat) - min(flat)) / min(flat)), flush=True)

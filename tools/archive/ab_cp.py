#!/usr/bin/env python3
"""In-process interleaved A/B of Chambolle-Pock variants on the north-star shape (the knobs go through
tv_set_option).  usage: python tools/ab_cp.py NAME=ENV1=V1,ENV2=V2 NAME2=... [--rounds 4] [--shape ...] [--scheme hybrid]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import torch, pytv
from bench import synth_slab
args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
shape = tuple(int(v) for v in sys.argv[sys.argv.index("--shape") + 1].split("x")) if "--shape" in sys.argv else (256, 8, 1024, 1024)
scheme = sys.argv[sys.argv.index("--scheme") + 1] if "--scheme" in sys.argv else "hybrid"
variants = []
for a in args:
    name, rest = a.split("=", 1)
    env = dict(kv.split("=") for kv in rest.split(",") if kv)
    variants.append((name, env))
x0 = synth_slab(shape, 0, shape[0], torch.device("cuda", 0))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        for k, v in env.items():
            if k.startswith("TV_"):
                pytv._native.set_option(k, int(v))
        fused = None if env.get("FUSED", "1") == "1" else False
        cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=scheme, reg_time=1.0 if shape[1] > 1 else 0.0, fused=fused)
        for _ in range(2):
            cp.step()
        cp.timing = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6):
            cp.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
        k1 = np.mean([e[0].elapsed_time(e[1]) for e in cp.timing]); k2 = np.mean([e[1].elapsed_time(e[2]) for e in cp.timing])
        res[name].append((dt * 1e3, k1, k2))
        for k in env:
            if k.startswith("TV_"):
                pytv._native.set_option(k, None)
        del cp
        torch.cuda.empty_cache()
for name, _ in variants:
    a = np.array(res[name])
    print("%-14s iter ms: median %.2f min %.2f | kernel1 median %.2f min %.2f | kernel2 median %.2f" % (
        name, np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), a[:, 1].min(), np.median(a[:, 2])))

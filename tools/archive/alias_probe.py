#!/usr/bin/env python3
"""Round 4, verdict item 1(c): does the FRAME PITCH of the solver's state decide the speed of the one-sweep CP iteration?

The ~20 streams a block of k_cp_fused touches per frame (8 q channels read + written, x, x0, p, x_out) are whole frames apart: with
dense 1024 x 1024 fp32 frames every one of them is congruent modulo 4 MiB.  This probe runs the SAME iteration in ONE process on
state with different frame pads (tv_geom::frame_pitch, ABI 4), interleaved and repeated, and checks that the padded runs
give bit-identical iterates.

usage: python tools/alias_probe.py [--shape 256x8x1024x1024] [--pads 0,4352,...] [--rounds 2] [--steps 8] [--scheme hybrid]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, pytv
from bench import synth_slab

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256x8x1024x1024")
ap.add_argument("--pads", default="0,4352,8448,16384,37120,65536,70400,0")
ap.add_argument("--rowpads", default="0")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--scheme", default="hybrid")
ap.add_argument("--check", type=int, default=1)
args = ap.parse_args()
shape = tuple(int(v) for v in args.shape.split("x"))
dev = torch.device("cuda", 0)
x0 = synth_slab(shape, 0, shape[0], dev)
ref_x = None
rows = []
for rnd in range(args.rounds):
    for rowpad in [int(v) for v in args.rowpads.split(",")]:
        for pad in [int(v) for v in args.pads.split(",")]:
            rp = shape[3] + rowpad // 4
            pitch = None if (pad == 0 and rowpad == 0) else (rp, shape[2] * rp + pad // 4)
            cp = pytv.solvers.ChambollePock(x0, 25.0, scheme=args.scheme, reg_time=1.0, pitch=pitch, fused=True)
            for _ in range(args.warmup):
                cp.step()
            cp.timing = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                cp.step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
            k1 = [e[0].elapsed_time(e[1]) for e in cp.timing]; k2 = [e[1].elapsed_time(e[2]) for e in cp.timing]
            same = None
            if args.check and rnd == 0:
                if ref_x is None and pitch is None:
                    ref_x = cp.x.clone()
                elif ref_x is not None:
                    same = bool(torch.equal(cp.x, ref_x))
            row = {"round": rnd, "frame_pad_bytes": pad, "row_pad_bytes": rowpad, "ms_per_it": round(dt * 1e3, 3), "sweep_ms_median": round(float(np.median(k1)), 3),
                   "sweep_ms_min": round(min(k1), 3), "sweep_ms_max": round(max(k1), 3), "fixup_ms_median": round(float(np.median(k2)), 3),
                   "bit_identical_to_dense": same, "q_ptr": hex(cp.q.data_ptr()), "x_ptr": hex(cp.x.data_ptr())}
            rows.append(row)
            print(json.dumps(row), flush=True)
            del cp
            torch.cuda.empty_cache()
print("# summary (median sweep ms by frame pad):")
for rowpad in sorted({r["row_pad_bytes"] for r in rows}):
    for pad in sorted({r["frame_pad_bytes"] for r in rows}):
        v = [r["sweep_ms_median"] for r in rows if r["frame_pad_bytes"] == pad and r["row_pad_bytes"] == rowpad]
        w = [r["ms_per_it"] for r in rows if r["frame_pad_bytes"] == pad and r["row_pad_bytes"] == rowpad]
        print("# row pad %4d frame pad %6d B: sweep %s   iteration %s" % (rowpad, pad, " ".join("%.2f" % a for a in v), " ".join("%.2f" % a for a in w)))

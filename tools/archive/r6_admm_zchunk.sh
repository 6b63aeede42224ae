#!/bin/bash
# Round 6: what the z-chunk length of the one-sweep kernels costs on the configs[4] rank slab (32 planes; the slab rule keeps 4 chunks of 8 so
# that the halo exchange hides behind interior chunks).  TV_NS_ZCHUNK pins the x-solve's own chunks at what they are.
O=gpurun_out
for s in upwind hybrid; do
  for zc in 0 8 16 32; do
    TV_ZCHUNK=$zc TV_NS_ZCHUNK=32 python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --pmc off --no-cpu-baseline > $O/r6k_admm_${s}_zc$zc.json 2>> $O/r6k_err.txt
    python3 - <<PY
import json
d=json.loads(open("$O/r6k_admm_${s}_zc$zc.json").read().strip().splitlines()[-1])
r=d["roofline"]; rx=d["roofline_xsolve"]; rf=d["roofline_fixup"]
print("$s TV_ZCHUNK=$zc  ms/iter %.3f  sweep %.3f (%.3f)  fixup %.3f  xsolve %.3f" % (d["ms_per_step"], r["ms_per_launch"], r["frac"], rf["ms_per_launch"], rx["ms_per_outer_iteration"]), flush=True)
PY
  done
done

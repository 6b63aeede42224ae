#!/bin/bash
# Round 6: tv_nstream.hip compiled with -fno-slp-vectorize (no packed-fp32 pairs: the copies into aligned register pairs sat right behind the
# plane loads) against the product build, interleaved on one box.  usage: PYTV4D_LIB of the variant in $1
VAR=$1; O=gpurun_out
for rep in 1 2; do
  for s in upwind hybrid central; do
    for v in product variant; do
      if [ $v = variant ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
      python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --pmc off --no-cpu-baseline > $O/r6i_admm_${s}_${v}_$rep.json 2>> $O/r6i_err.txt
      python3 - <<PY
import json
d=json.loads(open("$O/r6i_admm_${s}_${v}_$rep.json").read().strip().splitlines()[-1])
rx=d.get("roofline_xsolve",{})
print("$s $v rep $rep ms/iter %.3f xsolve frac %.4f loss %r" % (d["ms_per_step"], rx.get("frac"), d.get("loss_first_last")), flush=True)
PY
    done
  done
done
for v in product variant; do
  if [ $v = variant ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
  echo "== op_bench $v"; python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind central 2>/dev/null | grep -i "cheb\|normal"
done

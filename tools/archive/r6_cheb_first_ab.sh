#!/bin/bash
# Round 6: the first Chebyshev step as its own instantiation (CHEB = 2: no operand loads) against the general form (TV_NS_NO_FIRST=1),
# interleaved on one box: the ADMM line of bench.py on the configs[4] slab, and the kernel averages of the upwind run under rocprofv3.
O=gpurun_out; R=$(pwd)
for rep in 1 2; do
  for s in upwind hybrid central; do
    for nf in 0 1; do
      TV_NS_NO_FIRST=$nf python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --pmc off --no-cpu-baseline > $O/r6g_admm_${s}_nofirst${nf}_$rep.json 2>> $O/r6g_err.txt
      python3 - <<PY
import json
d=json.loads(open("$O/r6g_admm_${s}_nofirst${nf}_$rep.json").read().strip().splitlines()[-1])
rx=d.get("roofline_xsolve",{})
print("$s", "general" if $nf else "first-step form", "rep $rep", "ms/iter", d["ms_per_step"], "xsolve ms", rx.get("ms"), "frac", rx.get("frac"), flush=True)
PY
    done
  done
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6g_trace -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $R/$O/r6g_trace.json 2> $R/$O/r6g_trace.log )
grep -h "k_normal_stream" $(find $O/r6g_trace -name "*kernel_stats.csv" | head -1) | cut -c1-200
rm -rf $O/r6g_trace

#!/bin/bash
# EXPERIMENT -DTV_FUSED_PFQ=2: the dual channels of the next frame requested at the top of the current one for EVERY Nd = 4 scheme (the product
# does it for central only); one box, product library against libpytv4d_hip_pfq2.so (TV_VARIANT=pfq2 TV_EXTRA_FLAGS=-DTV_FUSED_PFQ=2 build.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
VAR=$R/pytv-4d_amd/pytv/libpytv4d_hip_pfq2.so
for rep in 1 2; do
for lib in product pfq2; do
  if [ $lib = pfq2 ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
  for s in upwind downwind; do
    python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/tmp_pfq.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_pfq.json').read().splitlines() if l.startswith('{')][-1]); print('$lib admm $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), d['loss_first_last'])"
    python3 bench.py --scheme $s --steps 12 --warmup 4 --no-cpu-baseline --pmc off --tune-placement off > $O/tmp_pfq.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_pfq.json').read().splitlines() if l.startswith('{')][-1]); print('$lib cp   $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3))"
  done
done
done

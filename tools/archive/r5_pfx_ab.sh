#!/bin/bash
# EXPERIMENT -DTV_FUSED_PFX=1: the operands of the lagged primal update (x0, p) requested at the top of the frame that finalises them; one box,
# product library against libpytv4d_hip_pfx.so (TV_VARIANT=pfx TV_EXTRA_FLAGS=-DTV_FUSED_PFX=1 build.py), two repetitions, interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
VAR=$R/pytv-4d_amd/pytv/libpytv4d_hip_pfx.so
PYTV4D_LIB=$VAR timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_admm_fused.py tests/test_gpu_cp_r4.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for lib in product pfx; do
  if [ $lib = pfx ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
  for s in hybrid upwind; do
    python3 bench.py --scheme $s --steps 12 --warmup 4 --no-cpu-baseline --pmc off --tune-placement off > $O/tmp_pfx.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_pfx.json').read().splitlines() if l.startswith('{')][-1]); print('$lib cp   $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), d.get('loss_first_last'))"
    python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/tmp_pfx.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_pfx.json').read().splitlines() if l.startswith('{')][-1]); print('$lib admm $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), d['loss_first_last'])"
  done
done
done

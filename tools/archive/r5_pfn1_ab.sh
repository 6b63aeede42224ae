#!/bin/bash
# EXPERIMENT -DTV_FUSED_PFN1=1: hybrid / downwind request x(z + 1) one frame ahead (tv_fused.h); one box: the parity tests with the variant, then
# tools/archive/slab_placement_probe.py with the product library and libpytv4d_hip_pfn1.so interleaved (4 constructions x 2 layouts x 3 repetitions each),
# then the ADMM / CP bench lines that use the kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
VAR=$R/pytv-4d_amd/pytv/libpytv4d_hip_pfn1.so
PYTV4D_LIB=$VAR timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_admm_fused.py tests/test_gpu_cp_r4.py tests/test_gpu_pitch.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in product pfn1; do
    if [ $lib = pfn1 ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
    echo "== $lib (rep $rep)"; python3 tools/archive/slab_placement_probe.py 256x8x1024x1024 4 slab+32,separate 2>&1 | grep -E "^slab|^separate" | cut -c1-200
  done
done
for rep in 1 2; do
for lib in product pfn1; do
  if [ $lib = pfn1 ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
  for s in hybrid downwind; do
    python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/tmp_p.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_p.json').read().splitlines() if l.startswith('{')][-1]); print('$lib admm $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), d['loss_first_last'])"
  done
  python3 bench.py --scheme downwind --steps 12 --warmup 4 --no-cpu-baseline --pmc off --tune-placement off > $O/tmp_p.json 2>/dev/null
  python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_p.json').read().splitlines() if l.startswith('{')][-1]); print('$lib cp   downwind', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3))"
done
done

#!/bin/bash
# round 5, late: the Chebyshev step / normal operator after the straight-line rewrite (frame descriptors, operands a frame ahead, planes loaded in
# place), A/B against the library of the commit before on ONE box:  PYTV4D_LIB=.../libpytv4d_hip_base.so (built from a worktree of that commit, or a copy of the product library of the commit before)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_admm_fused.py tests/test_gpu_admm_ops.py tests/test_gpu_parity.py tests/test_gpu_pitch.py tests/test_gpu_configs.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -5
BASE=$R/pytv-4d_amd/pytv/libpytv4d_hip_base.so
for rep in 1 2; do
for lib in new base; do
  [ $lib = base ] && [ ! -f $BASE ] && continue
  for s in upwind downwind central hybrid; do
    if [ $lib = base ]; then export PYTV4D_LIB=$BASE; else unset PYTV4D_LIB; fi
    python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/r5d_bench_admm_config4slab_${s}_${lib}.json 2>> $O/r5d_bench_admm_err.txt
    python3 -c "
import json; d=json.loads([l for l in open('$O/r5d_bench_admm_config4slab_${s}_${lib}.json').read().splitlines() if l.startswith('{')][-1]); print('$lib $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), 'xsolve', round(d['roofline_xsolve']['ms_per_outer_iteration'],3), round(d['roofline_xsolve']['frac'],3), d['loss_first_last'])"
  done
done
done
unset PYTV4D_LIB
echo "--- op_bench new"; python3 tools/op_bench.py 64x8x1024x1024 2>&1 | grep -i -E "cheb|normal" | head -20
if [ -f $BASE ]; then echo "--- op_bench base"; PYTV4D_LIB=$BASE python3 tools/op_bench.py 64x8x1024x1024 2>&1 | grep -i -E "cheb|normal" | head -20; fi
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5d_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/r5d_bench_admm_under_rocprof_trace.json 2> $O/r5d_trace_admm.log )
head -8 $(find $O/r5d_trace_admm -name "*kernel_stats.csv" | head -1) > $O/r5d_admm_config4slab_upwind_kernel_stats.csv
rm -rf $O/r5d_trace_admm
cut -c1-200 $O/r5d_admm_config4slab_upwind_kernel_stats.csv | head -5

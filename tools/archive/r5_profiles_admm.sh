#!/bin/bash
# round 5, late: the ADMM lines after the u ping-pong and the shared x / b stream of the first Chebyshev step (one gpurun call)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_admm_fused.py tests/test_gpu_admm_ops.py -x -q 2>&1 | tail -3
for s in upwind downwind central hybrid; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 > $O/r5c_bench_admm_config4slab_$s.json 2>> $O/r5c_bench_admm_err.txt
  python3 -c "
import json; d=json.loads([l for l in open('$O/r5c_bench_admm_config4slab_$s.json').read().splitlines() if l.startswith('{')][-1]); print('$s', d['ms_per_step'], 'sweep', d['roofline']['ms_per_launch'], round(d['roofline']['frac'],3), 'xsolve', d['roofline_xsolve']['ms_per_outer_iteration'], round(d['roofline_xsolve']['frac'],3), 'fixup', d['roofline_fixup']['ms_per_launch'], d['loss_first_last'])"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5c_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/r5c_bench_admm_under_rocprof_trace.json 2> $O/r5c_trace_admm.log )
head -14 $(find $O/r5c_trace_admm -name "*kernel_stats.csv" | head -1) > $O/r5c_admm_config4slab_upwind_kernel_stats.csv
rm -rf $O/r5c_trace_admm
cut -c1-200 $O/r5c_admm_config4slab_upwind_kernel_stats.csv | head -8

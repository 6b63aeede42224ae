#!/bin/bash
# after the streaming-normal-operator rewrite: the whole GPU suite, then the four ADMM lines + the default (CP north star) line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
python3 bench.py > $O/r5e_bench_default_first_command.json 2> $O/r5e_bench_default_err.txt
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5e_bench_default_first_command.json').read().splitlines() if l.startswith('{')][-1])
print('default bench: ms_per_step', d['ms_per_step'], 'it/s', d['value'], 'sweep', d['roofline']['ms_per_launch'], d['roofline']['frac'])
PY
for s in upwind downwind central hybrid; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 > $O/r5e_bench_admm_config4slab_$s.json 2>> $O/r5e_bench_admm_err.txt
  python3 -c "
import json; d=json.loads([l for l in open('$O/r5e_bench_admm_config4slab_$s.json').read().splitlines() if l.startswith('{')][-1]); print('$s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), round(d['roofline']['frac'],3), 'xsolve', round(d['roofline_xsolve']['ms_per_outer_iteration'],3), round(d['roofline_xsolve']['frac'],3), 'fixup', round(d['roofline_fixup']['ms_per_launch'],3), d['loss_first_last'])"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5e_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/r5e_bench_admm_under_rocprof_trace.json 2> $O/r5e_trace_admm.log )
head -8 $(find $O/r5e_trace_admm -name "*kernel_stats.csv" | head -1) > $O/r5e_admm_config4slab_upwind_kernel_stats.csv
rm -rf $O/r5e_trace_admm
cut -c1-200 $O/r5e_admm_config4slab_upwind_kernel_stats.csv | head -4
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r5e_fullsuite.txt; cat $O/r5e_fullsuite.txt

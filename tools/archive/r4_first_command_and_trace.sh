#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call14
mkdir -p "$O"
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_first.json 2> $O/bench_first.err
bash tools/prof.sh r4b_northstar --steps 20 --warmup 5 --tune-placement off > $O/prof.log 2>&1
python3 tools/pmc_digest.py $R/gpurun_out/prof_r4b_northstar $O r4b_fused_northstar >> $O/prof.log 2>&1
grep -h "^{" $R/gpurun_out/prof_r4b_northstar/trace.log | tail -1 > $O/bench_under_rocprof_trace.json
for i in 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_$i.json 2>/dev/null; done
python3 - $O/bench_first.json $O/bench_under_rocprof_trace.json $O/bench_2.json $O/bench_3.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d['series_ms']['kernel1']
        print(f.split('/')[-1], 'ms/step %.3f sweep %.3f (frac %.3f) fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline_fixup']['ms_per_launch']), 'min %.2f med %.2f max %.2f' % (s['min'], s['median'], s['max']), json.dumps(d.get('placement_tuning'))[:500])
    except Exception as e:
        print(f, 'unreadable', e)
PY
head -6 $O/r4b_fused_northstar_kernel_stats.csv | cut -c1-200
timeout 600 python3 -m pytest tests/test_gpu_cp_r4.py -x -q -m gpu 2>&1 | tail -3

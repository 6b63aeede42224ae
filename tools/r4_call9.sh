#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call9
mkdir -p "$O"
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_cp_r4.py -x -q -m gpu > $O/pytest_cp_r4.txt 2>&1; tail -15 $O/pytest_cp_r4.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off"
for i in 1 2 3; do
  $B > $O/bench_lazy_$i.json 2> $O/bench_lazy_$i.err
  TV_BENCH_LAZY=0 $B > $O/bench_classic_$i.json 2> $O/bench_classic_$i.err
done
python3 - $O/bench_lazy_1.json $O/bench_classic_1.json $O/bench_lazy_2.json $O/bench_classic_2.json $O/bench_lazy_3.json $O/bench_classic_3.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d['series_ms']['kernel1']
        print(f.split('/')[-1], 'ms/step %.3f sweep %.3f fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_fixup']['ms_per_launch']), 'min %.2f med %.2f max %.2f' % (s['min'], s['median'], s['max']), d['config'].get('kernels'), json.dumps(d.get('placement_tuning'))[:200])
    except Exception as e:
        print(f, 'unreadable', e, open(f.replace('.json','.err')).read()[-600:])
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call8
mkdir -p "$O"
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off"
for i in 1 2 3; do
  $B --tune-placement on > $O/bench_tune_$i.json 2> $O/bench_tune_$i.err
  $B --tune-placement off > $O/bench_notune_$i.json 2> $O/bench_notune_$i.err
done
python3 - $O/bench_tune_1.json $O/bench_notune_1.json $O/bench_tune_2.json $O/bench_notune_2.json $O/bench_tune_3.json $O/bench_notune_3.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d['series_ms']['kernel1']
        print(f.split('/')[-1], 'ms/step %.3f sweep %.3f fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_fixup']['ms_per_launch']), 'min %.2f med %.2f max %.2f' % (s['min'], s['median'], s['max']), json.dumps(d.get('placement_tuning'))[:600])
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 900 python3 tools/op_bench.py 64x8x1024x1024 > $O/op_bench_f32.txt 2>&1; tail -45 $O/op_bench_f32.txt | cut -c1-200
DTYPE=f64 timeout 900 python3 tools/op_bench.py 32x8x1024x1024 > $O/op_bench_f64.txt 2>&1; tail -45 $O/op_bench_f64.txt | cut -c1-200

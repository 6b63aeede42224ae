#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call6
mkdir -p "$O"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_1.json 2> $O/bench_1.err; python3 -c "
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.3f sweep %.3f fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_fixup']['ms_per_launch']), d['series_ms']['kernel1']['first5'])
" $O/bench_1.json
timeout 900 python3 -m pytest tests/test_gpu_cp_r4.py tests/test_gpu_weight_volume.py tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_1.txt 2>&1; tail -12 $O/pytest_1.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_2.json 2> $O/bench_2.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_3.json 2> $O/bench_3.err
python3 -c "
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.3f sweep %.3f fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_fixup']['ms_per_launch']), d['series_ms']['kernel1']['first5'])
" $O/bench_2.json $O/bench_3.json
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -25 $O/pytest_all.txt

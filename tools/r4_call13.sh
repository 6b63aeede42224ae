#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call13
mkdir -p "$O"
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_first.json 2> $O/bench_first.err; python3 -c "
import json;d=json.loads(open('$O/bench_first.json').read().strip().splitlines()[-1]);print('first command: ms/step %.3f sweep %.3f frac %.3f fixup %.3f traffic %s %s' % (d['ms_per_step'],d['roofline']['ms_per_launch'],d['roofline']['frac'],d['roofline_fixup']['ms_per_launch'],d['roofline']['traffic'],d['roofline_fixup']['traffic']))"
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python3 tools/stress_fused.py > $O/stress_fused.txt 2>&1; tail -2 $O/stress_fused.txt
timeout 900 python3 tools/stress_subgrad.py > $O/stress_subgrad.txt 2>&1; tail -2 $O/stress_subgrad.txt
timeout 900 python3 tools/stress_multirank.py > $O/stress_multirank.txt 2>&1; tail -2 $O/stress_multirank.txt
timeout 900 python3 tools/determinism_northstar.py > $O/determinism.txt 2>&1; tail -3 $O/determinism.txt
timeout 900 python3 tools/big_volume_check.py > $O/big_volume.txt 2>&1; tail -3 $O/big_volume.txt
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -5 $O/pytest_all.txt

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call10
mkdir -p "$O"
bash tools/r4_profiles.sh > $O/profiles.log 2>&1
python3 - gpurun_out/r4p/bench_northstar_first_command.json gpurun_out/r4p/bench_northstar_second.json gpurun_out/r4p/bench_northstar_no_tuner.json gpurun_out/r4p/bench_config1.json gpurun_out/r4p/bench_config2.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); s=d['series_ms']['kernel1']
        print(f.split('/')[-1], 'ms/step %.3f sweep %.3f (frac %.3f) fixup %.3f traffic %s / %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline_fixup']['ms_per_launch'], d['roofline'].get('traffic'), d['roofline_fixup'].get('traffic')), json.dumps(d.get('placement_tuning'))[:300])
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -15 $O/pytest_all.txt

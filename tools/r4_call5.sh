#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call5
mkdir -p "$O"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_first.json 2> $O/bench_first.err; cut -c1-400 $O/bench_first.json
timeout 600 tools/bwtest4 128 5 pingpong > $O/bwtest4_pingpong.txt 2>&1; grep -E "var 0|var 4" $O/bwtest4_pingpong.txt | cut -c1-200
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -25 $O/pytest_all.txt
timeout 900 python3 tools/stress_ops.py > $O/stress_ops.txt 2>&1; tail -5 $O/stress_ops.txt

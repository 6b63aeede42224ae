#!/bin/bash
# usage (on a FRESH GPU box): bash tools/placement_ab.sh "<bench args of run 1>" ; three runs back to back, run 1 with the given args
R=$GRAFT_REPO_ROOT
one() { python3 $R/bench.py --pmc off --no-cpu-baseline --steps 10 --warmup 3 $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('args [%s] ms/step %.3f sweep %.3f fixup %.3f' % ('$1',d['ms_per_step'],d['roofline']['ms_per_launch'],d['roofline_fixup']['ms_per_launch']))"; }
one "$1"; one ""; one ""

#!/bin/bash
# round 5, final library: the measurement pass behind DESIGN.md section 4 (ONE gpurun call; the first command of the lease is the driver-style
# bench line), then smoke() and the whole GPU suite.  Files: gpurun_out/r5g_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
python3 bench.py > $O/r5g_bench_default_first_command.json 2> $O/r5g_bench_err.txt
for s in upwind downwind central hybrid; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 > $O/r5g_bench_admm_config4slab_$s.json 2>> $O/r5g_bench_err.txt
done
for s in upwind downwind central; do
  python3 bench.py --scheme $s --steps 12 --warmup 4 --no-cpu-baseline > $O/r5g_bench_northstar_$s.json 2>> $O/r5g_bench_err.txt
done
python3 bench.py --workload config1 --steps 50 --warmup 10 --no-cpu-baseline > $O/r5g_bench_config1.json 2>> $O/r5g_bench_err.txt
python3 bench.py --workload config2 --steps 30 --warmup 5 --no-cpu-baseline > $O/r5g_bench_config2.json 2>> $O/r5g_bench_err.txt
python3 bench.py --workload config3 --allow-single --steps 6 --warmup 2 --no-cpu-baseline --pmc off > $O/r5g_bench_config3_single_gpu.json 2>> $O/r5g_bench_err.txt
python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central > $O/r5g_op_rooflines.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5g_trace_cp -o t -- python3 $R/bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline --tune-placement off > $O/r5g_bench_northstar_under_rocprof_trace.json 2> $O/r5g_trace_cp.log )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5g_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/r5g_bench_admm_under_rocprof_trace.json 2> $O/r5g_trace_admm.log )
head -12 $(find $O/r5g_trace_cp -name "*kernel_stats.csv" | head -1) > $O/r5g_fused_northstar_kernel_stats.csv
head -12 $(find $O/r5g_trace_admm -name "*kernel_stats.csv" | head -1) > $O/r5g_admm_config4slab_upwind_kernel_stats.csv
rm -rf $O/r5g_trace_cp $O/r5g_trace_admm
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
for f in sorted(glob.glob(O + "/r5g_bench_*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = d.get("roofline") or {}
    x = d.get("roofline_xsolve") or {}
    print(os.path.basename(f), "ms", round(d.get("ms_per_step") or 0, 3), "value", round(d.get("value") or 0, 3), "sweep", round(r.get("ms_per_launch") or 0, 3), round(r.get("frac") or 0, 3),
          "xsolve", round(x.get("ms_per_outer_iteration") or 0, 3), round(x.get("frac") or 0, 3))
PY
cut -c1-160 $O/r5g_admm_config4slab_upwind_kernel_stats.csv | head -4; cut -c1-160 $O/r5g_fused_northstar_kernel_stats.csv | head -4
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r5g_fullsuite.txt; cat $O/r5g_fullsuite.txt

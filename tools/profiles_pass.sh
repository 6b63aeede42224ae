#!/bin/bash
# The measurement pass behind DESIGN.md section 4 as ONE gpurun call: bash tools/profiles_pass.sh <tag>   (files: gpurun_out/<tag>_*)
# The first command of the lease is the driver-style bench line; then the ADMM line for the four schemes, the other BASELINE configs, the
# operator rooflines, the descent loop, the reference's own small shapes, rocprofv3 kernel traces of the same commands and the PMC passes.
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
python3 bench.py > $O/${TAG}_bench_default_first_command.json 2> $O/${TAG}_bench_err.txt
for s in upwind downwind central hybrid; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 > $O/${TAG}_bench_admm_config4slab_$s.json 2>> $O/${TAG}_bench_err.txt
done
for s in upwind downwind central; do
  python3 bench.py --scheme $s --steps 12 --warmup 4 --no-cpu-baseline > $O/${TAG}_bench_northstar_$s.json 2>> $O/${TAG}_bench_err.txt
done
python3 bench.py --workload config1 --steps 50 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_config1.json 2>> $O/${TAG}_bench_err.txt
python3 bench.py --workload config2 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_config2.json 2>> $O/${TAG}_bench_err.txt
python3 bench.py --workload config3 --allow-single --steps 6 --warmup 2 --no-cpu-baseline --pmc off > $O/${TAG}_bench_config3_single_gpu.json 2>> $O/${TAG}_bench_err.txt
python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central > $O/${TAG}_op_rooflines.txt 2>&1
SG_TUNE=0 python3 tools/sg_bench.py 256x8x1024x1024 hybrid upwind downwind central > $O/${TAG}_sg_loop_northstar.txt 2>&1
python3 tools/small_volume_bench.py --schemes hybrid,upwind,downwind,central 20x4x100x100 1x1x512x512 1x1x256x256 20x1x100x100 2>&1 | grep -v amdgpu.ids > $O/${TAG}_small_volume_bench.txt
# rocprofv3 kernel traces of the SAME commands (tuner off: exactly the run's launches)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_cp -o t -- python3 $R/bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline --tune-placement off > $O/${TAG}_bench_northstar_under_rocprof_trace.json 2> $O/${TAG}_trace_cp.log )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/${TAG}_bench_admm_under_rocprof_trace.json 2> $O/${TAG}_trace_admm.log )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_small -o t -- python3 $R/tools/small_volume_bench.py --schemes hybrid 20x4x100x100 > $O/${TAG}_small_volume_under_rocprof_trace.txt 2> $O/${TAG}_trace_small.log )
head -12 $(find $O/${TAG}_trace_cp -name "*kernel_stats.csv" | head -1) > $O/${TAG}_fused_northstar_kernel_stats.csv
head -16 $(find $O/${TAG}_trace_admm -name "*kernel_stats.csv" | head -1) > $O/${TAG}_admm_config4slab_upwind_kernel_stats.csv
head -12 $(find $O/${TAG}_trace_small -name "*kernel_stats.csv" | head -1) > $O/${TAG}_small_volume_kernel_stats.csv
rm -rf $O/${TAG}_trace_cp $O/${TAG}_trace_admm $O/${TAG}_trace_small
# PMC passes of the north-star line (FETCH_SIZE / WRITE_SIZE / L2 in separate runs) -> digest
bash tools/prof.sh ${TAG}pmc --tune-placement off > $O/${TAG}_prof.log 2>&1
mkdir -p $O/${TAG}_digest && python3 tools/pmc_digest.py $O/prof_${TAG}pmc $O/${TAG}_digest ${TAG}_fused_northstar > /dev/null 2>&1
cp $O/${TAG}_digest/${TAG}_fused_northstar_pmc_digest.json $O/ 2>/dev/null
rm -rf $O/prof_${TAG}pmc $O/${TAG}_digest
python3 - "$TAG" <<'PY'
import json, glob, os, sys
tag = sys.argv[1]
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
for f in sorted(glob.glob(O + "/%s_bench_*.json" % tag)):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = d.get("roofline") or {}
    x = d.get("roofline_xsolve") or {}
    print(os.path.basename(f), "ms", round(d.get("ms_per_step") or 0, 3), "value", round(d.get("value") or 0, 3), "sweep", round(r.get("ms_per_launch") or 0, 3), round(r.get("frac") or 0, 3),
          "xsolve", round(x.get("ms_per_outer_iteration") or 0, 3), round(x.get("frac") or 0, 3))
PY
cut -c1-160 $O/${TAG}_fused_northstar_kernel_stats.csv | head -4; cut -c1-160 $O/${TAG}_small_volume_kernel_stats.csv | head -5
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2

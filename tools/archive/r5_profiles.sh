#!/bin/bash
# round 5: the measurement pass behind DESIGN.md section 4 (run as ONE gpurun call; the first command of the lease is the driver-style bench line)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench_northstar_first_command.json 2> $O/r5_bench_northstar_err.txt
for s in upwind downwind central hybrid; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 > $O/r5_bench_admm_config4slab_$s.json 2>> $O/r5_bench_admm_err.txt
done
python3 bench.py --workload config1 --steps 50 --warmup 10 > $O/r5_bench_config1.json 2>> $O/r5_bench_northstar_err.txt
python3 bench.py --workload config2 --steps 30 --warmup 5 > $O/r5_bench_config2.json 2>> $O/r5_bench_northstar_err.txt
python3 bench.py --workload config3 --allow-single --steps 6 --warmup 2 --no-cpu-baseline --pmc off > $O/r5_bench_config3_single_gpu.json 2>> $O/r5_bench_northstar_err.txt
python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central > $O/r5_op_rooflines.txt 2>&1
SG_TUNE=0 python3 tools/sg_bench.py 256x8x1024x1024 hybrid upwind downwind central > $O/r5_sg_loop_northstar.txt 2>&1
# rocprofv3 kernel trace of the SAME bench command (tuner off: exactly the run's launches), then of the ADMM line
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_trace_cp -o t -- python3 $R/bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline --tune-placement off > $O/r5_bench_northstar_under_rocprof_trace.json 2> $O/r5_trace_cp.log )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_trace_admm -o t -- python3 $R/bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 --pmc off --no-cpu-baseline --tune-placement off > $O/r5_bench_admm_under_rocprof_trace.json 2> $O/r5_trace_admm.log )
head -12 $(find $O/r5_trace_cp -name "*kernel_stats.csv" | head -1) > $O/r5_fused_northstar_kernel_stats.csv
head -16 $(find $O/r5_trace_admm -name "*kernel_stats.csv" | head -1) > $O/r5_admm_config4slab_upwind_kernel_stats.csv
rm -rf $O/r5_trace_cp $O/r5_trace_admm
# counters of the aligned-tile sub-gradient kernel (upwind) and of the round-3 tile (hybrid): traffic + SQ breakdown
OPS=tv_subgrad_fused bash tools/prof_op.sh r5_sg_upwind 64x8x1024x1024 upwind > $O/r5_prof_sg_upwind.log 2>&1
cp $O/op_r5_sg_upwind/digest.json $O/r5_subgrad_aligned_upwind_pmc_digest.json 2>/dev/null
rm -rf $O/op_r5_sg_upwind
for f in r5_bench_northstar_first_command r5_bench_admm_config4slab_upwind r5_bench_admm_config4slab_hybrid r5_bench_config3_single_gpu; do echo "== $f"; head -c 600 $O/$f.json; echo; done
cat $O/r5_fused_northstar_kernel_stats.csv | cut -c1-200; cat $O/r5_op_rooflines.txt | head -60; cat $O/r5_sg_loop_northstar.txt

#!/bin/bash
# round-4 evidence run (on the GPU box): bash tools/r4_profiles.sh  -> gpurun_out/r4p/*
# The FIRST command is the bench exactly as the driver runs it (live PMC traffic, CPU baselines).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_northstar_first_command.json 2> $O/bench_northstar.err
bash tools/prof.sh r4_northstar --steps 10 --warmup 3 > $O/prof_northstar.log 2>&1
python3 tools/pmc_digest.py $R/gpurun_out/prof_r4_northstar $O r4_fused_northstar >> $O/prof_northstar.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/bench_northstar_second.json 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc off --tune-placement off > $O/bench_northstar_no_tuner.json 2>/dev/null
for w in config1 config2; do python3 bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$w.json 2>/dev/null; done
python3 tools/op_bench.py 64x8x1024x1024 > $O/op_rooflines_f32.txt 2>&1
DTYPE=f64 python3 tools/op_bench.py 32x8x1024x1024 > $O/op_rooflines_f64.txt 2>&1
python3 tools/sg_bench.py 256x8x1024x1024 > $O/sg_loop_northstar.txt 2>&1
python3 tools/admm_bench.py 32x16x1024x1024 5 > $O/admm_config4_slab.txt 2>&1
ls -la $O

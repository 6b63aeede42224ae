#!/bin/bash
# round-3 evidence run (on the GPU box): bash tools/r3_profiles.sh  -> gpurun_out/r3p/*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 3 > $O/bench_northstar.json 2> $O/bench_northstar.err
bash tools/prof.sh r3_northstar --steps 10 --warmup 3 > $O/prof_northstar.log 2>&1
python3 tools/pmc_digest.py $R/gpurun_out/prof_r3_northstar $O r3_fused_northstar >> $O/prof_northstar.log 2>&1
python3 tools/op_bench.py 64x8x1024x1024 > $O/op_rooflines_f32.txt 2>&1
DTYPE=f64 python3 tools/op_bench.py 32x8x1024x1024 > $O/op_rooflines_f64.txt 2>&1
python3 tools/sg_bench.py 256x8x1024x1024 > $O/sg_loop_northstar.txt 2>&1
for s in hybrid upwind central; do
  OPS=tv_subgrad_fused bash tools/prof_op.sh r3_sg_$s 64x8x1024x1024 $s > $O/sg_prof_$s.txt 2>&1
  cp $R/gpurun_out/op_r3_sg_$s/digest.json $O/sg_${s}_digest.json
  find $R/gpurun_out/op_r3_sg_$s/trace -name "*kernel_stats.csv" -exec cp {} $O/sg_${s}_kernel_stats.csv \;
done
DTYPE=f64 OPS=tv_subgrad_fused bash tools/prof_op.sh r3_sg_f64 32x8x1024x1024 hybrid > $O/sg_prof_f64.txt 2>&1
cp $R/gpurun_out/op_r3_sg_f64/digest.json $O/sg_f64_hybrid_digest.json
for w in config1 config2; do python3 bench.py --workload $w --pmc off --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$w.json 2>/dev/null; done
python3 tools/admm_bench.py 32x16x1024x1024 5 > $O/admm_config4_slab.txt 2>&1
ONLY=one-sweep bash tools/prof_admm.sh r3_fused 32x16x1024x1024 5 > $O/admm_prof.txt 2>&1
cp $R/gpurun_out/admm_r3_fused/kernel_stats_top.csv $O/admm_one_sweep_kernel_stats.csv; cp $R/gpurun_out/admm_r3_fused/digest.json $O/admm_one_sweep_pmc_digest.json
ls -la $O

#!/bin/bash
# usage (on the GPU box): bash tools/prof.sh <tag> [bench args...]
# kernel trace + stats, then two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --pmc off --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --pmc off --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/bench.py --pmc off --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o l -- python3 $R/bench.py --pmc off --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -20

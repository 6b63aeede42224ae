#!/bin/bash
# usage (on the GPU box): [VARIANT=ea] [BENCH_ARGS="--scheme upwind"] [SCRIPT="tools/op_bench.py 256x8x1024x1024 hybrid" KFILTER=stream] bash tools/fetch_ab.sh
# FETCH_SIZE of the Chambolle-Pock kernels with the default library and with a variant build (TV_VARIANT=... build.py)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fetch_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base ${VARIANT:-ea}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v -o f -- python3 $R/${SCRIPT:-bench.py --pmc off --no-cpu-baseline --steps 2 --warmup 1} $BENCH_ARGS > $OUT/$v.log 2>&1
  python3 - $OUT/$v $v "${KFILTER:-cp_f}" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if sys.argv[3] in k: print(sys.argv[2], k, "reads 2xFETCH = %.2f GB (%d launches)" % (2 * sum(v) / len(v) * 1024 / 1e9, len(v)))
PY
  rm -rf $OUT/$v
done

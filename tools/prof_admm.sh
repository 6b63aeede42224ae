#!/bin/bash
# usage (on the GPU box): bash tools/prof_admm.sh <tag> [NzxMxNyxNx] [n_cg]
# rocprofv3 kernel trace + stats and two PMC passes (FETCH_SIZE / WRITE_SIZE) over tools/admm_bench.py
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/admm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/admm_bench.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/tools/admm_bench.py "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/tools/admm_bench.py "$@" > $OUT/pmc_write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "tv::" in k and "k_reduce" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dig = {}
for k, d in agg.items():
    e = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_GB_per_launch"] = {"read_2xFETCH": 2 * e["FETCH_SIZE"]["mean"] * 1024 / 1e9, "write": e["WRITE_SIZE"]["mean"] * 1024 / 1e9}
    dig[k] = e
json.dump(dig, open(out + "/digest.json", "w"), indent=1)
PY
head -24 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) > $OUT/kernel_stats_top.csv
cat $OUT/kernel_stats_top.csv | cut -c1-160

#!/bin/bash
# usage (on the GPU box): OPS=tv_subgrad_fused bash tools/prof_op.sh <tag> <shape> <scheme...>
# kernel trace + separate PMC passes over tools/op_bench.py (one operator selected with OPS=...)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/op_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/op_bench.py "$@" > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- python3 $R/tools/op_bench.py "$@" > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, collections, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "tv::" in k and "k_reduce" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dig = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in dig.items():
    if "SQ_WAVE_CYCLES" in d:
        wc = d["SQ_WAVE_CYCLES"]
        d["frac_wave_cycles"] = {n: d[c] / wc for n, c in (("wait_any", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                                                          ("issuing", "SQ_ACTIVE_INST_ANY")) if c in d}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_GB"] = {"read_2xFETCH": 2 * d["FETCH_SIZE"] * 1024 / 1e9, "write": d["WRITE_SIZE"] * 1024 / 1e9}
json.dump(dig, open(out + "/digest.json", "w"), indent=1)
print(json.dumps(dig, indent=1))
PY
head -8 $OUT/trace/*/t_kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*stats*" | head

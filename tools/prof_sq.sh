#!/bin/bash
# usage (on the GPU box): bash tools/prof_sq.sh <tag> [bench args...]   -- SQ wave-cycle breakdown of the kernels of a bench.py run
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- python3 $R/bench.py --no-cpu-baseline --pmc off --steps 2 --warmup 1 "$@" > $OUT/pmc$i.log 2>&1
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
        d["frac_wave_cycles"] = {n: round(d[c] / wc, 3) for n, c in (("wait_any", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                                                                     ("issuing", "SQ_ACTIVE_INST_ANY")) if c in d}
        if "SQ_WAVES" in d:
            d["per_wave"] = {c: round(d[c] / d["SQ_WAVES"], 1) for c in d if c.startswith("SQ_INSTS")}
json.dump(dig, open(out + "/digest.json", "w"), indent=1)
for k, d in dig.items():
    print(k[:60], d.get("frac_wave_cycles"), d.get("per_wave"))
PY

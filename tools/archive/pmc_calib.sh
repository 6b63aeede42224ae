#!/bin/bash
# usage (on the GPU box): bash tools/pmc_calib.sh [nz]
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels whose HBM bytes are known exactly (tools/bwtest2: every array is
# read / written once, nothing is shared between threads), for the access shapes the TV kernels use (16 B per lane, row
# segments of 64 B ... 1 KiB per wave).  Prints counter bytes / exact bytes per kernel.
NZ=${1:-64}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_calib
mkdir -p $OUT
[ -x $R/tools/bwtest2 ] || hipcc -O3 --offload-arch=gfx950 $R/tools/bwtest2.hip -o $R/tools/bwtest2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- $R/tools/bwtest2 $NZ > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" "$NZ" <<'PY'
import csv, collections, glob, json, re, sys
out, nz = sys.argv[1], int(sys.argv[2])
V = nz * 8 * 1024 * 1024
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = {}
for k, d in sorted(agg.items()):
    m = re.match(r"k_(\w+?)<(.*)>", k)
    if not m: continue
    kind, targs = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if kind == "copy": rd = wr = V * 8 * 4 / 2
    elif kind in ("march", "march_tile"): rd, wr = (1 + 8) * 4 * V, 8 * 4 * V
    elif kind == "dstore": rd, wr = 4 * V, int(targs[1]) * 4 * V
    elif kind == "dtload": rd, wr = 8 * 4 * V, 4 * V
    else: continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    rows[k] = {"exact_read_GB": rd / 1e9, "exact_write_GB": wr / 1e9, **c}
    if "FETCH_SIZE" in c: rows[k]["2xFETCH_over_exact"] = 2 * c["FETCH_SIZE"] * 1024 / rd
    if "WRITE_SIZE" in c: rows[k]["WRITE_over_exact"] = c["WRITE_SIZE"] * 1024 / wr
json.dump(rows, open(out + "/calib.json", "w"), indent=1)
for k, r in rows.items():
    print("%-34s read 2xFETCH/exact %.3f   WRITE/exact %.3f   hit %.3g miss %.3g" % (k, r.get("2xFETCH_over_exact", -1), r.get("WRITE_over_exact", -1),
          r.get("TCC_HIT_sum", -1), r.get("TCC_MISS_sum", -1)))
PY

#!/usr/bin/env python3
"""Digest a tools/prof.sh output directory (gpurun_out/prof_<tag>) into the small files kept under
profiles/: the rocprofv3 kernel-stats CSV (top rows) and per-kernel averages of the PMC passes.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, so the read bytes are 2 x FETCH_SIZE (MI355X_MICROARCH.md, section HBM)."""
import collections
import csv
import json
import os
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
rows = list(csv.reader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
    csv.writer(f).writerows(rows[:12])
digest = collections.OrderedDict()
for name, path in (("FETCH_SIZE", "pmc_fetch/f_counter_collection.csv"), ("WRITE_SIZE", "pmc_write/w_counter_collection.csv"),
                   ("L2", "pmc_l2/l_counter_collection.csv")):
    p = os.path.join(src, path)
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "tv::" in r["Kernel_Name"] and "k_reduce" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        digest.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
for k, d in digest.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd = 2.0 * d["FETCH_SIZE"]["mean"] * 1024.0
        wr = d["WRITE_SIZE"]["mean"] * 1024.0
        d["hbm_bytes_per_launch"] = {"read_2xFETCH": rd, "write": wr, "total": rd + wr}
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        h, m = d["TCC_HIT_sum"]["mean"], d["TCC_MISS_sum"]["mean"]
        d["l2_hit_rate"] = h / (h + m)
json.dump(digest, open(os.path.join(dst, "%s_pmc_digest.json" % tag), "w"), indent=1)
print(json.dumps(digest, indent=1))

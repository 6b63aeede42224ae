#!/usr/bin/env python3
"""Per-instance (L2 channel x XCD) values of the counters of a `rocprofv3 --pmc ... --output-format json` run, for the k_cp_fused
dispatches: the CSV output sums over DIMENSION_INSTANCE[0:15] x DIMENSION_XCC[0:7]; the JSON keeps one record per instance.
Prints, per dispatch of the sweep: duration, and for every counter the sum / min / max / max-over-mean of the per-instance values,
and LEVEL / REQ (mean latency in L2 cycles) per instance spread where both counters are present.
usage: python tools/pmc_channels.py <rocprofv3 output dir>"""
import glob, json, os, sys

d = sys.argv[1]
files = sorted(glob.glob(os.path.join(d, "**", "*results.json"), recursive=True)) or sorted(glob.glob(os.path.join(d, "**", "*.json"), recursive=True))
if not files:
    print("no json under", d)
    sys.exit(0)
doc = json.load(open(files[0]))
root = doc.get("rocprofiler-sdk-tool", doc)
root = root[0] if isinstance(root, list) else root
print("# file", files[0], "top-level keys:", list(root.keys()))
cb = root.get("callback_records", {})
print("# callback_records keys:", list(cb.keys()) if isinstance(cb, dict) else type(cb))
cc = cb.get("counter_collection", []) if isinstance(cb, dict) else []
print("# counter_collection entries:", len(cc))
if cc:
    print("# first entry (truncated):", json.dumps(cc[0])[:1500])
# counter id -> name
names = {}
for c in root.get("counters", []) or []:
    try:
        names[c["id"]["handle"]] = c["name"]
    except Exception:
        pass
ksym = {}
for k in root.get("kernel_symbols", []) or []:
    try:
        ksym[k["kernel_id"]] = k.get("formatted_kernel_name") or k.get("kernel_name")
    except Exception:
        pass
print("# counters known:", len(names), " kernel symbols:", len(ksym))
n = 0
for e in cc:
    try:
        dd = e.get("dispatch_data", {})
        info = dd.get("dispatch_info", {})
        kname = ksym.get(info.get("kernel_id"), str(info.get("kernel_id")))
        if "k_cp_fused" not in str(kname):
            continue
        per = {}
        for r in e.get("records", []):
            cid = r.get("counter_id", {}).get("handle")
            per.setdefault(names.get(cid, str(cid)), []).append(float(r.get("value", 0.0)))
        t0, t1 = dd.get("start_timestamp"), dd.get("end_timestamp")
        dur = (t1 - t0) * 1e-6 if (t0 and t1) else float("nan")
        line = "dispatch %s  %.3f ms" % (info.get("dispatch_id"), dur)
        for name, v in sorted(per.items()):
            m = sum(v) / len(v)
            line += " | %s n=%d sum %.4g min %.4g max %.4g max/mean %.3f" % (name, len(v), sum(v), min(v), max(v), max(v) / m if m else 0)
        lv = [k for k in per if k.endswith("_LEVEL")]
        for k in lv:
            rq = k[:-6]
            if rq in per and len(per[rq]) == len(per[k]):
                lat = [a / b for a, b in zip(per[k], per[rq]) if b > 0]
                if lat:
                    lat.sort()
                    line += " | %s/REQ cycles: min %.0f median %.0f max %.0f" % (k, lat[0], lat[len(lat) // 2], lat[-1])
        print(line)
        n += 1
    except Exception as exc:
        print("# entry failed:", exc)
print("# sweep dispatches:", n)

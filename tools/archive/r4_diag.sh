#!/bin/bash
# Round 4, verdict item 1: why is the FIRST north-star run on a fresh box 9 % slower than profiles/ ?
# Run as the first and only command of a gpurun lease:   bash tools/r4_diag.sh
# Everything lands in gpurun_out/r4diag/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4diag
mkdir -p "$O"
export TMPDIR=/tmp
B="python3 bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline"
digest() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("series_ms", {})
    print("%-28s ms/step %.3f sweep %.3f fixup %.3f | sweep first5 %s last5 %s | clocks before %s after %s | layout %s" % (
        sys.argv[1].split("/")[-1], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline_fixup"]["ms_per_launch"],
        s.get("kernel1", {}).get("first5"), s.get("kernel1", {}).get("last5"),
        {k: v for k, v in (d["gpu_state"]["before"] or {}).items() if k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_junction_c", "temp_mem_c")},
        {k: v for k, v in (d["gpu_state"]["after"] or {}).items() if k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_junction_c", "temp_mem_c")},
        d.get("state_layout", {}).get("frame_pad_bytes")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
date +"%s start" > $O/timeline.txt
# ---- (a) the first command of the lease: the bench alone, dense state -----------------------------------------------
$B --pitch none > $O/bench_1_first_dense.json 2> $O/bench_1.err; digest $O/bench_1_first_dense.json | tee -a $O/summary.txt
date +"%s after first bench" >> $O/timeline.txt
# ---- (b) the same with a 0.2 Hz rocm-smi poller beside it (what the driver does) -------------------------------------
( while true; do rocm-smi --showuse --showpower --showclocks --showmeminfo vram --json > $O/smi.$(date +%s).json 2>/dev/null; sleep 5; done ) &
POLL=$!
sleep 1
$B --pitch none > $O/bench_2_poller_dense.json 2> $O/bench_2.err; digest $O/bench_2_poller_dense.json | tee -a $O/summary.txt
kill $POLL 2>/dev/null; wait $POLL 2>/dev/null
# ---- (c) padded state (frame pitch 4 MiB + 4352 B), alone, then dense again, then padded again ------------------------
$B --pitch 4352 > $O/bench_3_pad4352.json 2> $O/bench_3.err; digest $O/bench_3_pad4352.json | tee -a $O/summary.txt
$B --pitch none > $O/bench_4_dense.json 2> $O/bench_4.err; digest $O/bench_4_dense.json | tee -a $O/summary.txt
$B --pitch 4352 > $O/bench_5_pad4352.json 2> $O/bench_5.err; digest $O/bench_5_pad4352.json | tee -a $O/summary.txt
date +"%s after benches" >> $O/timeline.txt
# ---- (d) the sweep's memory shape without arithmetic, pitches as arguments -------------------------------------------
timeout 600 tools/bwtest4 128 5 > $O/bwtest4.txt 2>&1; grep -c GB/s $O/bwtest4.txt
date +"%s after bwtest4" >> $O/timeline.txt
# ---- (e) the real iteration, one process, interleaved frame pads, bit-identity check ---------------------------------
timeout 900 python3 tools/alias_probe.py --rounds 2 --steps 8 > $O/alias_probe.txt 2> $O/alias_probe.err; tail -12 $O/alias_probe.txt
date +"%s after alias probe" >> $O/timeline.txt
# ---- (f) parity of the pitched path -------------------------------------------------------------------------------------
timeout 900 python3 -m pytest tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_pitch.txt 2>&1; tail -3 $O/pytest_pitch.txt
# ---- (g) counters: what is there, and memory-side latency / per-channel requests of a dense and a padded run ----------
rocprofv3 --list-avail > $O/avail.txt 2>&1
P="bench.py --steps 2 --warmup 1 --pmc off --no-cpu-baseline"
for L in none 4352; do
  for C in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
    T=$(echo $C | tr ' ' '+')
    D=/tmp/r4pmc_${L}_$T
    rm -rf $D
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -o p -- python3 $R/$P --pitch $L > $O/pmc_${L}_$T.log 2>&1 )
    if [ "$T" = "TCC_EA0_RDREQ+TCC_EA0_WRREQ" ]; then find $D -name "*counter_collection.csv" -exec sh -c 'head -1 "$1"; grep k_cp_fused "$1" | head -400' _ {} \; > $O/pmc_raw_${L}.csv; fi
    python3 - $D "$L $C" >> $O/pmc_summary.txt <<'PY'
import csv, glob, os, sys
rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_cp_fused" in k or "k_cp_fixup<3, 0" in k:
            rows.setdefault((k.split("(")[0][:60], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
print("##", sys.argv[2])
for (k, c), v in sorted(rows.items()):
    print("  %-62s %-34s n=%d mean %.6g" % (k, c, len(v), sum(v) / len(v)))
PY
  done
done
date +"%s end" >> $O/timeline.txt
cat $O/summary.txt

#!/bin/bash
# per-kernel device durations of the decoder variants (rocprofv3 --kernel-trace --stats), one process per variant
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out/decode_prof_${1:-r02}.txt
: > $OUT
rm -rf gpurun_out/dprof_*
for V in "64 0 0" "64 1 0" "64 0 1" "64 1 1" "1 0 0" "1 1 0" "1 1 1" "16 0 1 stress" "16 1 1 stress" "16 1 0 stress"; do
  TAG=$(echo $V | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dprof_$TAG -- python3 tools/decode_prof.py $V > gpurun_out/dprof_$TAG.log 2>&1
  F=$(find gpurun_out/dprof_$TAG -name '*kernel_stats.csv' | head -1)
  T=$(find gpurun_out/dprof_$TAG -name '*kernel_trace.csv' | head -1)
  echo "== B fused exact: $V" >> $OUT
  python3 - "$F" "$T" >> $OUT <<'PY'
import csv, sys
import numpy as np
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").split("(")[0]
    if (n.startswith("sd::") and "render" not in n) or "fill" in n:
        print(f"   {n[:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
# device span of one decode call (first kernel start -> last kernel end, i.e. including the gaps between dependent launches)
NAMES = ("k_nms_tile", "k_nms_slots", "k_select_group", "k_select_map", "k_select_peaks", "fillBuffer", "k_decode_fused", "k_map_stream_select", "k_rank_maps", "k_group_wide")
rows = [r for r in csv.DictReader(open(sys.argv[2])) if any(k in r["Kernel_Name"] for k in NAMES)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# launches per decode call: the kernel sequence repeats, so the period is the distance between two launches of the first kernel's name
first = rows[0]["Kernel_Name"]
per = next((i for i in range(1, len(rows)) if rows[i]["Kernel_Name"] == first), len(rows))
spans = [(int(rows[i + per - 1]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in range(0, len(rows) - per + 1, per)][10:]
print(f"   launches per call {per}; device span per call (incl. launch gaps): median {np.median(spans):.2f} us, min {min(spans):.2f} us")
PY
done
cat $OUT

#!/bin/bash
# per-kernel device durations of sd_decode on the stress geometry (BASELINE configs[4], bs = 16): launch pair with one selector block per
# image against the map-parallel path at both tile heights, exact top-k and annotations-only (rocprofv3 --kernel-trace --stats)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out/decode_prof_stress_${1:-r04}.txt
: > $OUT
for V in "1073741824 0 1 0" "1 32 1 0" "1 0 1 1" "1 0 0 1" "1073741824 0 0 0"; do
  set -- $V
  TAG=from$1_th$2_exact$3_stream$4
  rm -rf gpurun_out/dps_$TAG
  SD_MAP_FROM=$1 SD_MAP_TH=$2 SD_MAP_STREAM=$4 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dps_$TAG -- python3 tools/decode_prof.py 16 0 $3 stress > gpurun_out/dps_$TAG.log 2>&1
  F=$(find gpurun_out/dps_$TAG -name '*kernel_stats.csv' | head -1)
  T=$(find gpurun_out/dps_$TAG -name '*kernel_trace.csv' | head -1)
  echo "== map_parallel_from $1, map_tile_height $2, exact_topk $3, map_stream $4 (bs 16, 1024x1024, 8 + 8 maps, K 128, P 512)" >> $OUT
  python3 - "$F" "$T" >> $OUT <<'PY'
import csv, sys
import numpy as np
names = ("k_nms_tile", "k_select_group", "fillBuffer", "k_nms_slots", "k_select_map", "k_rank_maps", "k_group_wide", "k_map_stream_select")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").split("(")[0]
    if any(k in n for k in names):
        print(f"   {n[:44]:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
rows = [r for r in csv.DictReader(open(sys.argv[2])) if any(k in r["Kernel_Name"] for k in names)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len({r["Kernel_Name"].split("(")[0] for r in rows})
spans = [(int(rows[i + per - 1]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in range(0, len(rows) - per + 1, per)][10:]
print(f"   device span per call (incl. launch gaps): median {np.median(spans):.2f} us = {np.median(spans) / 16:.2f} us per image, min {min(spans):.2f} us")
PY
done
cat $OUT

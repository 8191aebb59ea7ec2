#!/bin/bash
# round 5: the map-parallel decoder with its maps split over 1 .. 4 blocks of the streaming kernel and with / without the one-launch
# rank + association kernel (rocprofv3 --kernel-trace --stats, one process per variant).  usage: tools/decode_prof_r05.sh [tag]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out/decode_split_${1:-r05}.txt
: > $OUT
rm -rf gpurun_out/dsplit_*
#            B fused exact [stress] | split rank_group
#            B fused exact [stress] | split rank_group half
DEFAULT_VARIANTS="64 0 0|1 0 0;64 0 0|1 1 0;64 0 0|1 1 1;64 0 0|2 1 1;64 0 1|1 0 0;64 0 1|1 1 1;512 0 0|1 0 0;512 0 0|3 1 0;512 0 0|0 1 1;512 0 0|1 1 1;16 0 1 stress|1 0 0;16 0 0 stress|1 0 0"
IFS=';' read -ra VARIANTS <<< "${SD_DECODE_VARIANTS:-$DEFAULT_VARIANTS}"      # (override: SD_DECODE_VARIANTS="16 0 1 stress|2 0 0;...")
for V in "${VARIANTS[@]}"; do
  ARGS=${V%%|*}; OPT=${V##*|}
  set -- $OPT
  export SD_MAP_SPLIT=$1 SD_MAP_RG=$2 SD_MAP_HALF=$3
  TAG=$(echo "$ARGS $OPT" | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dsplit_$TAG -- python3 tools/decode_prof.py $ARGS > gpurun_out/dsplit_$TAG.log 2>&1
  F=$(find gpurun_out/dsplit_$TAG -name '*kernel_stats.csv' | head -1)
  T=$(find gpurun_out/dsplit_$TAG -name '*kernel_trace.csv' | head -1)
  echo "== B fused exact: $ARGS | map_split $SD_MAP_SPLIT map_rank_group $SD_MAP_RG map_half $SD_MAP_HALF" >> $OUT
  python3 - "$F" "$T" >> $OUT <<'PY'
import csv, sys
import numpy as np
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").split("(")[0]
    if (n.startswith("sd::") and "render" not in n) or "fill" in n:
        print(f"   {n[:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
NAMES = ("k_nms_tile", "k_nms_slots", "k_select_group", "k_select_map", "k_select_peaks", "fillBuffer", "k_decode_fused", "k_map_stream_select", "k_rank_maps",
         "k_group_wide", "k_rank_group")
rows = [r for r in csv.DictReader(open(sys.argv[2])) if any(k in r["Kernel_Name"] for k in NAMES)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = rows[0]["Kernel_Name"]
per = next((i for i in range(1, len(rows)) if rows[i]["Kernel_Name"] == first), len(rows))
spans = [(int(rows[i + per - 1]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in range(0, len(rows) - per + 1, per)][10:]
print(f"   launches per call {per}; device span per call (incl. launch gaps): median {np.median(spans):.2f} us, min {min(spans):.2f} us")
PY
done
rm -rf gpurun_out/dsplit_*/
cat $OUT

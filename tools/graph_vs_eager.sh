#!/bin/bash
# bs=1 fp32 forward, eager launches vs one hipGraph replay, under rocprofv3 --kernel-trace: per-launch listing of the last forward of each
# (kernel time, idle time between kernels) -- where a replay spends the time an eager forward does not.  Output: gpurun_out/graph_vs_eager.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out/graph_vs_eager.txt
: > $OUT
for MODE in eager graph; do
  rm -rf gpurun_out/gve_$MODE
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gve_$MODE -- python3 tools/prof_fwd1.py 30 $MODE > gpurun_out/gve_$MODE.log 2>&1
  T=$(find gpurun_out/gve_$MODE -name '*kernel_trace.csv' | head -1)
  echo "== $MODE" >> $OUT
  python3 - "$T" >> $OUT <<'PY'
import csv, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_stem_fwd" in r["Kernel_Name"]]
spans, busys, gapss, ns = [], [], [], []
for a, b in zip(idx[5:-1], idx[6:]):
    # a forward = from its stem launch up to (not including) whatever precedes the next stem launch that is not part of it (the input copy)
    rep = rows[a:b]
    while rep and ("copy" in rep[-1]["Kernel_Name"].lower() or "elementwise" in rep[-1]["Kernel_Name"]): rep = rep[:-1]
    spans.append((int(rep[-1]["End_Timestamp"]) - int(rep[0]["Start_Timestamp"])) / 1e3)
    busys.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rep) / 1e3)
    gapss.append([(int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3 for x, y in zip(rep, rep[1:])])
    ns.append(len(rep))
    period = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    spans[-1] = (spans[-1], period)
print(f"   kernels per forward {int(np.median(ns))}; stem start -> head end {np.median([s[0] for s in spans]):.1f} us; period (stem start -> next stem start) {np.median([s[1] for s in spans]):.1f} us")
print(f"   kernel time {np.median(busys):.1f} us; idle between kernels {np.median([sum(g) for g in gapss]):.1f} us (median gap {np.median([x for g in gapss for x in g]):.2f} us, max {max(x for g in gapss for x in g):.2f} us)")
PY
done
cat $OUT

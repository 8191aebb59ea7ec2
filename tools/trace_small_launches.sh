#!/bin/bash
# kernel trace of tools/prof_train.py (4 steps): the launches of the LAST step in order, with durations and the gap to the previous launch
# usage: tools/trace_small_launches.sh [amp]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf gpurun_out/ktrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktrace -- python3 tools/prof_train.py 4 $1 > gpurun_out/ktrace.log 2>&1
T=$(find gpurun_out/ktrace -name '*kernel_trace.csv' | head -1)
python3 - "$T" > gpurun_out/step_launches_${1:-fp32}.txt <<'PY'
import csv, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
lo, hi = adam[-2] + 1, adam[-1] + 1          # the last step: after the previous step's Adam up to and including its own
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
print(f"# last step: {len(step)} launches, span {span:.1f} us, sum of kernel durations {busy:.1f} us, idle {span - busy:.1f} us")
c = Counter(); d = Counter()
for r in step:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    c[n] += 1; d[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("# launches per step / total us / name")
for n, k in sorted(c.items(), key=lambda x: -d[x[0]]):
    print(f"{k:5d} {d[n]:10.1f}  {n}")
print("# the step in order: start us, duration us, gap to the previous launch's end us, stream, name")
prev = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].replace('void ', '').split('(')[0][:70]}")
    prev = max(prev, e)
PY
head -60 gpurun_out/step_launches_${1:-fp32}.txt

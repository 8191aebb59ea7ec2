#!/bin/bash
# memory-copy trace of the training steps (which copies, how big): usage bash tools/prof_memcpy.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --memory-copy-trace --output-format csv -d gpurun_out/prof_memcpy -- python3 tools/prof_train.py 2 > gpurun_out/prof_memcpy.log 2>&1 || (tail -5 gpurun_out/prof_memcpy.log; exit 1)
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_memcpy/**/*memory_copy_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
agg = collections.Counter()
for r in rows:
    agg[(r.get("Direction"), r.get("Size") or r.get("Bytes"))] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:20]:
    print(k, v)
PY

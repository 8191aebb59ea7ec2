#!/bin/bash
# PMC counters for the conv micro-benchmark (own run, kernel-trace only).  usage: bash tools/prof_pmc.sh <tag> "<counters>" [conv_bench args]
set -e
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 tools/conv_bench.py --iters 1 "$@" > gpurun_out/pmc_$TAG.log 2>&1 || (tail -20 gpurun_out/pmc_$TAG.log; exit 1)
find gpurun_out/pmc_$TAG -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} gpurun_out/pmc_$TAG.csv
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc_$TAG.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY

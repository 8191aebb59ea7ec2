#!/bin/bash
# PMC counters for the bf16 conv micro-benchmark.  usage: bash tools/pmc_any_bf16.sh "<counters>" [conv_bench_bf16 args]
set -e
CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf gpurun_out/pmc_any16
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmc_any16 -- python3 tools/conv_bench_bf16.py --iters 1 --no-ab "$@" > gpurun_out/pmc_any16.log 2>&1 || (tail -5 gpurun_out/pmc_any16.log; exit 1)
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_any16/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    agg[r["Kernel_Name"].split("(")[0][-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "conv" in k:
        print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in d.items()})
PY

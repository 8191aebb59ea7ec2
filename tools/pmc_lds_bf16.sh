#!/bin/bash
# LDS bank conflicts of the bf16 conv kernels (args: passed to tools/conv_bench_bf16.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf gpurun_out/pmc_lds16
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d gpurun_out/pmc_lds16 -- python3 tools/conv_bench_bf16.py --iters 1 --no-ab "$@" > gpurun_out/pmc_lds16.log 2>&1 || (tail -5 gpurun_out/pmc_lds16.log; exit 1)
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_lds16/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    agg[r["Kernel_Name"].split("(")[0][-32:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "conv" in k:
        print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in d.items()})
PY

#!/bin/bash
# effective clock + MFMA busy of the bf16 conv kernels: GRBM_GUI_ACTIVE / 8 / duration  (args: passed to tools/conv_bench_bf16.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rm -rf gpurun_out/pmc_clk16
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_clk16 -- python3 tools/conv_bench_bf16.py --iters 2 "$@" > gpurun_out/pmc_clk16.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_clk16/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/pmc_clk16/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for d, v in vals.items():
    ns, name = dur.get(d, (0, "?"))
    if ("conv" in name) and ns > 20000:
        agg[(name.split("(")[0][-30:], round(ns / 5e3))].append((ns, v.get("GRBM_GUI_ACTIVE", 0) / 8 / ns, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 0)))
for (name, _), rows in sorted(agg.items()):
    ns = sum(r[0] for r in rows) / len(rows); ghz = sum(r[1] for r in rows) / len(rows)
    mf = sum(r[2] for r in rows) / len(rows); bc = sum(r[3] for r in rows) / len(rows)
    print(f"{name:32s} x{len(rows):3d} {ns/1e3:8.1f} us  clock {ghz:.3f} GHz  mfma_busy {mf:.3e} busy_cu {bc:.3e}  mfma/busy_cu {mf/max(bc,1):.3f}")
PY

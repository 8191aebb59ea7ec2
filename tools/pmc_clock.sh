#!/bin/bash
# effective clock + MFMA busy of the conv kernels: GRBM_GUI_ACTIVE / 8 / duration
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python3 tools/conv_bench.py --iters 1 --only fwd > gpurun_out/pmc_clk.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for d, v in vals.items():
    ns, name = dur.get(d, (0, "?"))
    if "igemm" in name and ns > 300000:
        rows.append((ns, name.split("(")[0][-24:], v.get("GRBM_GUI_ACTIVE", 0) / 8 / ns, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 0)))
for ns, name, ghz, mf, bc in sorted(rows)[-8:]:
    print(f"{name} {ns/1e3:8.1f} us  clock {ghz:.3f} GHz  mfma_busy {mf:.3e} busy_cu {bc:.3e}  mfma/busy_cu {mf/max(bc,1):.3f}")
PY

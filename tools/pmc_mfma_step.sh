#!/bin/bash
# MFMA utilisation of the kernels of the fp32 training step (bs=64, 512x512) from PMC counters, one pass, kernel-trace only:
# per kernel: effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CU_CYCLES.
# usage: bash tools/pmc_mfma_step.sh [script args...]   (default: tools/prof_train.py 2; e.g. tools/prof_train.py 2 amp, tools/prof_bf16_fwd.py 3)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 ${@:-tools/prof_train.py 2} > gpurun_out/pmc_mfma.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/pmc_mfma/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for d, v in vals.items():
    ns, name = dur.get(d, (0, "?"))
    k = name.split("(")[0].replace("void ", "").replace("sd::", "")
    a = agg[k]
    a[0] += 1; a[1] += ns; a[2] += v.get("GRBM_GUI_ACTIVE", 0.0); a[3] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[4] += v.get("SQ_BUSY_CU_CYCLES", 0.0)
print(f"{'kernel':44s} {'launches':>8s} {'avg us':>9s} {'clock GHz':>9s} {'MFMA busy / CU busy':>20s}")
for k, (n, ns, gui, mf, bc) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / n < 20e3 or mf == 0:
        continue
    print(f"{k[:44]:44s} {n:8d} {ns / n / 1e3:9.1f} {gui / 8 / ns:9.3f} {mf / max(bc, 1):20.3f}")
PY

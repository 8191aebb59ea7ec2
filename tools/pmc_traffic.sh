#!/bin/bash
# HBM traffic of the conv kernels from PMC counters: two separate passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled (gfx950 counts 128-B requests at 64 B).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_$C -- python3 tools/prof_train.py 2 > gpurun_out/pmc_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{C}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != C: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[C] = {"launches": n, "avg_kb": v / n}
res = {}
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and ("conv" in k or "stem" in k or "wgrad" in k):
        fetch = 2.0 * d["FETCH_SIZE"]["avg_kb"] * 1024      # gfx950 correction
        write = d["WRITE_SIZE"]["avg_kb"] * 1024
        res[k] = {"launches": d["FETCH_SIZE"]["launches"], "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": fetch + write}
import hashlib
res["_meta"] = {"sd_conv_hip_sha256": hashlib.sha256(open("structuredetector_amd/csrc/sd_conv.hip", "rb").read()).hexdigest(),
                "workload": "tools/prof_train.py 2 (bs=64, 512x512 fp32 training steps)", "counters": "FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, separate passes"}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in res.items():
    if k == "_meta": continue
    print(f"{k[:44]:44s} launches {v['launches']:4d}  read {v['hbm_read_bytes_per_launch']/1e6:9.1f} MB  write {v['hbm_write_bytes_per_launch']/1e6:8.1f} MB")
PY

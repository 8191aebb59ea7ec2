cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr_*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_train -- python3 tools/prof_train.py 3 > gpurun_out/tr_train.log 2>&1
T=$(find gpurun_out/tr_train -name '*kernel_trace.csv' | head -1)
echo "== copyBuffer neighbours (fp32 train step)"; python3 tools/trace_neighbours.py $T copyBuffer 1 | head -12
echo "== fillBuffer neighbours"; python3 tools/trace_neighbours.py $T fillBuffer 1 | head -8
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_eager -- python3 tools/prof_fwd1.py 40 > gpurun_out/tr_eager.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_graph -- python3 tools/prof_fwd1.py 40 graph > gpurun_out/tr_graph.log 2>&1
for m in eager graph; do T=$(find gpurun_out/tr_$m -name '*kernel_trace.csv' | head -1); echo "== bs=1 forward, $m"; python3 - $T <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels; last 50 names:", [r["Kernel_Name"].replace("void ","").split("(")[0][:22] for r in rows[-50:]][:50])
PY
done

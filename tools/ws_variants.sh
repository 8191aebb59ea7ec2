#!/bin/bash
# builds libsdnet_hip.so with different warp-specialised igemm parameters on the GPU box and times the forward convs
set -e
for V in "32 2" "16 3" "16 4"; do
  set -- $V
  make -C structuredetector_amd/csrc clean > /dev/null
  make -C structuredetector_amd/csrc EXTRA="-DSD_WS_BK=$1 -DSD_WS_STAGES=$2" > gpurun_out/mk_$1_$2.log 2>&1
  echo "=== BK=$1 stages=$2"
  timeout -k 10 200 python tools/conv_bench.py --only fwd 2>/dev/null | grep -E "l1.3x3|l2 3x3|l3 3x3|l4 3x3|up4.conv|total"
done

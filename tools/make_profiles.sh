#!/bin/bash
# Regenerates every profile artefact of a round in ONE gpurun call; results land in gpurun_out/profiles_<tag>/ (copy to profiles/).
# usage: bash tools/make_profiles.sh r02
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT gpurun_out/mp_* && mkdir -p $OUT
stats() {   # stats <name> <python script + args>: rocprofv3 --kernel-trace --stats of one program, kernel_stats csv kept
  local NAME=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mp_$NAME -- python3 "$@" > $OUT/${TAG}_${NAME}.stdout 2> gpurun_out/mp_$NAME.err
  cp "$(find gpurun_out/mp_$NAME -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_${NAME}.csv
}
# 1. the default bench command's timed region under rocprofv3 (side figures and CPU baseline off: per-kernel averages = training steps)
stats bench_no_extras bench.py --no-extras --no-cpu-baseline
grep '^{' $OUT/${TAG}_bench_no_extras.stdout > $OUT/${TAG}_bench_under_rocprof.json
# 2. the default bench line itself (no profiler)
python3 bench.py > $OUT/${TAG}_bench_default_run.json 2> gpurun_out/mp_bench.err
# 3. training steps only: fp32 and mixed precision; bf16 eval forward
stats train_fp32 tools/prof_train.py 4
stats train_amp tools/prof_train.py 4 amp
stats fwd_bf16 tools/prof_bf16_fwd.py 6
stats fwd_bf16_stress tools/prof_bf16_fwd.py 6 stress
stats fwd_bs1 tools/prof_fwd1.py 20
# 4. decoder variants (per-kernel durations + device span per call)
bash tools/decode_prof_r05.sh $TAG > /dev/null 2>&1 && cp gpurun_out/decode_split_$TAG.txt $OUT/${TAG}_decode_variants.txt
# 5. HBM traffic of the conv kernels (PMC, separate passes)
bash tools/pmc_traffic.sh > $OUT/${TAG}_pmc_traffic.stdout 2>&1 && cp gpurun_out/pmc_traffic.json $OUT/${TAG}_pmc_hbm_traffic_conv_kernels.json
ls -la $OUT

#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (run on the GPU box through gpurun).
# usage: bash tools/prof_kernel_trace.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/bench_prof_$TAG.log 2>&1
find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
head -25 gpurun_out/kernel_stats_$TAG.csv

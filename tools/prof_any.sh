#!/bin/bash
# usage: bash tools/prof_any.sh <tag> <python script + args>   -> rocprofv3 kernel stats
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 "$@" > gpurun_out/prof_$TAG.log 2>&1 || (tail -20 gpurun_out/prof_$TAG.log; exit 1)
find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
tail -3 gpurun_out/prof_$TAG.log
head -12 gpurun_out/kernel_stats_$TAG.csv | cut -c1-160

#!/bin/bash
# rocprofv3 kernel-trace summary of one bench run (run on the GPU box through gpurun).  Usage: tools/profile.sh <tag>
set -u
TAG=${1:-r01}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$TAG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline ) > gpurun_out/prof_$TAG/bench_stdout.log 2>&1
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${TAG}_kernel_stats.csv
rm -f gpurun_out/prof_$TAG/*.db gpurun_out/prof_$TAG/*kernel_trace.csv
ls -R gpurun_out/prof_$TAG | head -30

#!/bin/bash
# rocprofv3 kernel-trace summary of one bench run on the default workload (C3, end to end).  Run on the GPU box through gpurun.
# Usage: tools/profile.sh <tag> [extra bench.py arguments]
set -u
TAG=${1:-r02}
shift || true
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$TAG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --resident-steps 5 --no-cpu-baseline --no-dense --no-bwa "$@" ) > gpurun_out/prof_$TAG/bench_stdout.log 2>&1
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${TAG}_kernel_stats.csv
rm -f gpurun_out/prof_$TAG/*.db gpurun_out/prof_$TAG/*kernel_trace.csv
tail -c 400 gpurun_out/prof_$TAG/bench_stdout.log
head -30 gpurun_out/prof_${TAG}_kernel_stats.csv

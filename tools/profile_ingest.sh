#!/bin/bash
# rocprofv3 kernel-trace summary of one GPU ingest of workload C3 (run on the GPU box through gpurun)
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p /tmp/c3 gpurun_out/prof_ingest
build/gen_synth_bam --config C3 --out /tmp/c3/C3 --threads 32 > /dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_ingest -o trace -- python3 $REPO/tools/ingest_once.py /tmp/c3/C3 ) > gpurun_out/prof_ingest/stdout.log 2>&1
find gpurun_out/prof_ingest -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_ingest_kernel_stats.csv
rm -f gpurun_out/prof_ingest/*.db gpurun_out/prof_ingest/*kernel_trace.csv
tail -12 gpurun_out/prof_ingest/stdout.log
head -8 gpurun_out/prof_ingest_kernel_stats.csv | cut -c1-200

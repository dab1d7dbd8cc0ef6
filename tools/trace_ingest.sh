#!/bin/bash
# kernel timeline of GPU ingests of workload C3 (run on the GPU box through gpurun): which ingest kernels overlap, and for how long.
# Writes gpurun_out/ingest_timeline_<tag>.txt (a per-kernel-name summary + busy intervals) from rocprofv3's kernel trace.
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p /tmp/c3 gpurun_out/trace_$TAG
[ -f /tmp/c3/C3.bam ] || build/gen_synth_bam --config C3 --out /tmp/c3/C3 --threads 64 > /dev/null
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/trace_$TAG -o trace -- python3 $REPO/tools/ingest_twice.py /tmp/c3/C3 ) > gpurun_out/trace_$TAG/stdout.log 2>&1
T=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_summary.py "$T" --list > gpurun_out/ingest_timeline_$TAG.txt
rm -f gpurun_out/trace_$TAG/*.db "$T"
tail -3 gpurun_out/trace_$TAG/stdout.log
head -20 gpurun_out/ingest_timeline_$TAG.txt

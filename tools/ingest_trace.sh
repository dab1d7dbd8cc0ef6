#!/bin/bash
# rocprofv3 kernel trace of staged C3 steps (tools/staged_steps.py) + the summary of tools/ingest_trace.py.  GPU box only.
# Usage: tools/ingest_trace.sh <prefix of the C3 BAM pair> <tag>   (environment switches are passed through)
set -u
PRE=$1; TAG=$2
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_$TAG
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/trace_$TAG -o t -- python3 $REPO/tools/staged_steps.py $PRE ) > gpurun_out/trace_$TAG/stdout.log 2>&1
grep "^== " gpurun_out/trace_$TAG/stdout.log | tail -3
F=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/ingest_trace.py $F
rm -rf gpurun_out/trace_$TAG

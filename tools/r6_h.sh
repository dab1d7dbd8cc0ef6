#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6h /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
tools/ingest_trace.sh /tmp/squid_bench/C3 cur > gpurun_out/r6h/trace.log 2>&1
head -22 gpurun_out/r6h/trace.log | cut -c1-260
grep "launches:" gpurun_out/r6h/trace.log | cut -c1-700

#!/bin/bash
# kernel trace of three staged C3 steps; tools/ingest_trace.py over the last of them
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
REPO=$PWD
mkdir -p /tmp/squid_bench gpurun_out/r6t
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
ls /tmp/squid_bench | head
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6t -o tr -- python3 $REPO/tools/staged_steps.py /tmp/squid_bench/C3 3 ) > gpurun_out/r6t/run.log 2>&1
F=$(find /tmp/r6t -name "*kernel_trace.csv" | head -1)
python3 tools/ingest_trace.py "$F" > gpurun_out/r6t/trace.txt 2>&1
cut -c1-1500 gpurun_out/r6t/trace.txt
tail -3 gpurun_out/r6t/run.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
REPO=$PWD
mkdir -p /tmp/squid_bench gpurun_out/r6tf
[ -f /tmp/squid_bench/C3_s20180003.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3_s20180003 --threads 32 > /dev/null 2>&1
rm -rf /tmp/r6tf
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/r6tf -o tr -- python3 $REPO/tools/file_steps.py /tmp/squid_bench/C3_s20180003 4 ) > gpurun_out/r6tf/run.log 2>&1
grep "== step" gpurun_out/r6tf/run.log
KT=$(find /tmp/r6tf -name "*kernel_trace.csv" | head -1); MT=$(find /tmp/r6tf -name "*memory_copy_trace.csv" | head -1)
head -2 $MT | cut -c1-300
python3 tools/file_trace.py "$KT" "$MT" > gpurun_out/r6tf/trace.txt 2>&1
cut -c1-2500 gpurun_out/r6tf/trace.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6tl /tmp/squid_bench
[ -f /tmp/squid_bench/C3_s20180003.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3_s20180003 --threads 32 > /dev/null 2>&1
python3 tools/file_step_timeline.py > gpurun_out/r6tl/tl.txt 2>&1
grep -E "== step|== staged|copy threads|last piece|pieces queued|file streamed|queued at|bytes left" gpurun_out/r6tl/tl.txt | cut -c1-260 | tail -40

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6c /tmp/squid_bench
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 38410 35210 41610 44810 38409 38410 2>&1 | grep variant | tee gpurun_out/r6c/tok_bench2.log

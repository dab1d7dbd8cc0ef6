#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6f /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for rt in 1 2 4; do
  SQUID_RESOLVE_T=$rt python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 25610 2>&1 | grep variant | sed "s/^/T=$rt /" | tee -a gpurun_out/r6f/resolve.log
done

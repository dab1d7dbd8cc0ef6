#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6o2 /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for ring in 0 1; do
  SQUID_RESOLVE_RING=$ring python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 25610 2>&1 | grep variant | sed "s/^/ring=$ring /" | tee -a gpurun_out/r6o2/resolve.log
done

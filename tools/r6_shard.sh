#!/bin/bash
# shard projection (tools/shard_project.py) with the final reader, at several batch sizes for the ranks' short ranges
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6sh /tmp/squid_bench
[ -f /tmp/squid_bench/C3_s20180003.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3_s20180003 --threads 32 > /dev/null 2>&1
for cap in default 128 256 512; do
  if [ $cap = default ]; then unset SQUID_TOK_CAP_MB; else export SQUID_TOK_CAP_MB=$cap; fi
  echo "== SQUID_TOK_CAP_MB=$cap"
  timeout 600 python3 tools/shard_project.py /tmp/squid_bench/C3_s20180003 2 8 2>&1 | grep -E "unsharded|^W=|ingest per rank" | cut -c1-250
done | tee gpurun_out/r6sh/shard.txt

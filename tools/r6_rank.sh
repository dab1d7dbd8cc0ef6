#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6rk /tmp/squid_bench
[ -f /tmp/squid_bench/C3_s20180003.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3_s20180003 --threads 32 > /dev/null 2>&1
for r in 3 7; do python3 tools/rank_ingest.py /tmp/squid_bench/C3_s20180003 8 $r 3 > gpurun_out/r6rk/rank$r.txt 2>&1; grep -E "^shard|== rank|GPU ingest|BAI|chimeric|host" gpurun_out/r6rk/rank$r.txt | tail -32 | cut -c1-260; done

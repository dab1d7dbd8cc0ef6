#!/bin/bash
# from-file C3 steps under a few settings of the streamed read (piece size, copies in flight, reader threads)
cd "$(dirname "$0")/.."
[ -f /tmp/squid_bench/C3_s20180003.bam ] || { mkdir -p /tmp/squid_bench; build/gen_synth_bam --config C3 --out /tmp/squid_bench/C3_s20180003 --threads 128 > /dev/null; }
for cfg in "A=1" "SQUID_FEED_PIECE_MB=16" "SQUID_FEED_PIECE_MB=4" "SQUID_FEED_INFLIGHT=16" "SQUID_FEED_INFLIGHT=4" "SQUID_FEED_PIECE_MB=16,SQUID_FEED_INFLIGHT=16" "SQUID_FEED_THREADS=32" "SQUID_FEED_THREADS=8"; do
  echo "== $cfg"
  env ${cfg//,/ } python3 tools/file_step_timeline.py 2>&1 | grep -E "^== step|last [0-9]" | head -8 | tr '\n' ' '; echo
done

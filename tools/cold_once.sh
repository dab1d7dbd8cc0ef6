#!/bin/bash
# cold `squid` on C3 three times (4 s of rest in front of each), the phase lines of SQUID_TIMING / SQUID_INGEST_TIMING
B=build; W=/tmp/sqprobe; mkdir -p $W
[ -f $W/c3.bam ] || $B/gen_synth_bam --config C3 --out $W/c3 --threads 64 > /dev/null
for d in 1 2 3; do sleep 4; echo "=== run $d"; SQUID_TIMING=1 SQUID_INGEST_TIMING=1 timeout 120 $B/squid -b $W/c3.bam -c $W/c3.chim.bam -o $W/out 2>&1 >/dev/null | grep "squid +\|first two batches\|batch 0 planned\|entry +\|ingest .*chim"; done

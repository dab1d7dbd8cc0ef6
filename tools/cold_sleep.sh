#!/bin/bash
# cold `squid` runs on C3 with the GPU left alone for a while in front of each (the driver wipes the VRAM a process releases)
O=gpurun_out/${1:-r04c}; mkdir -p $O
W=/tmp/squid_bench; B=build
pre=$(ls $W/C3_s20180003.bam 2>/dev/null | sed 's/.bam$//')
[ -z "$pre" ] && { mkdir -p /tmp/sqprobe; $B/gen_synth_bam --config C3 --out /tmp/sqprobe/c3 --threads 64 > /dev/null; pre=/tmp/sqprobe/c3; }
for s in 0 2 5 5; do
  sleep $s
  t0=$(date +%s.%N)
  SQUID_TIMING=1 SQUID_INGEST_TIMING=1 $B/squid -b $pre.bam -c $pre.chim.bam -o /tmp/cold_out > /dev/null 2> $O/cold_sleep${s}_$RANDOM.err
  t1=$(date +%s.%N)
  echo "sleep $s: $(echo "$t1 - $t0" | bc) s" >> $O/cold_sleep.txt
done
cat $O/cold_sleep.txt

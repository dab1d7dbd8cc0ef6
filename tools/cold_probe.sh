#!/bin/bash
# GPU box: where does a cold `squid` process on C3 spend its wall clock, and how fast can file bytes reach HBM?
set -u
O=gpurun_out/${1:-r04b}; mkdir -p $O
W=/tmp/sqprobe; mkdir -p $W
B=build
[ -f $W/c3.bam ] || $B/gen_synth_bam --config C3 --out $W/c3 --threads 64 > /dev/null
ls -la $W > $O/files.txt
for i in 1 2 3; do
  SQUID_TIMING=1 SQUID_INGEST_TIMING=1 $B/squid -b $W/c3.bam -c $W/c3.chim.bam -o $W/out > $O/cold_${i}.out 2> $O/cold_${i}.err
done
[ "${2:-}" = probe ] && [ -x $B/h2d_probe ] && $B/h2d_probe $W/c3.bam > $O/h2d_probe.txt 2>&1

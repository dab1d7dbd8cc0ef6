#!/bin/bash
# C5 (dense config) on the GPU box: where the time of a step goes.  Usage: tools/c5_probe.sh [records]
REC=${1:-30000000}
OUT=gpurun_out/c5_probe
mkdir -p $OUT /tmp/c5p
set -x
( time build/gen_synth_bam --config C5 --out /tmp/c5p/C5 --records $REC --threads 32 ) > $OUT/gen.log 2>&1
hipcc -O3 -std=c++17 -o /tmp/c5p/chim_probe tools/chim_probe.cpp squid_amd/csrc/sq_bam.cpp squid_amd/csrc/sq_chimeric.cpp -lz -lpthread -ldl > $OUT/probe_build.log 2>&1
SQUID_INGEST_TIMING=1 SQUID_CHIM_PROF=1 timeout 300 /tmp/c5p/chim_probe /tmp/c5p/C5.chim.bam 16 > $OUT/chim_probe.log 2>&1
( time SQUID_TIMING=1 SQUID_INGEST_TIMING=1 SQUID_CHIM_PROF=1 timeout 600 build/squid -b /tmp/c5p/C5.bam -c /tmp/c5p/C5.chim.bam -o /tmp/c5p/cli -w 1 -a 50 --threads 16 ) > $OUT/cli.log 2>&1
( time SQUID_TIMING=1 SQUID_REPLAY_PROF=1 SQUID_ORDER_PROF=1 timeout 600 build/squid -b /tmp/c5p/C5.bam -c /tmp/c5p/C5.chim.bam -o /tmp/c5p/cli2 -w 1 -a 50 --threads 16 ) > $OUT/cli_replay_prof.log 2>&1
cmp /tmp/c5p/cli_sv.txt /tmp/c5p/cli2_sv.txt && echo same_sv >> $OUT/cli.log
( time SQUID_TIMING=1 SQUID_HOST_FILTERS=1 timeout 600 build/squid -b /tmp/c5p/C5.bam -c /tmp/c5p/C5.chim.bam -o /tmp/c5p/cli3 -w 1 -a 50 --threads 16 ) > $OUT/cli_host_filters.log 2>&1
cmp /tmp/c5p/cli_sv.txt /tmp/c5p/cli3_sv.txt && echo same_sv >> $OUT/cli_host_filters.log
nproc > $OUT/nproc.log; cat /sys/fs/cgroup/cpu.max >> $OUT/nproc.log 2>&1

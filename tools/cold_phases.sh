#!/bin/bash
# cold `build/squid` runs with the phase clock, under a few settings of the reader's buffers: what does the exit of the process cost?
# usage: tools/cold_phases.sh [prefix]
cd "$(dirname "$0")/.."
PRE=${1:-/tmp/squid_bench/C3_s20180003}
[ -f $PRE.bam ] || { mkdir -p $(dirname $PRE); build/gen_synth_bam --config C3 --out $PRE --threads 128 > /dev/null; }
run() {
  sleep 4
  local t0=$(date +%s%N)
  env "$@" SQUID_PHASES=1 SQUID_T0_NS=$t0 build/squid -b $PRE.bam -c $PRE.chim.bam -o /tmp/cold_x > /dev/null 2> /tmp/cold_x.err
  local t1=$(date +%s%N)
  echo "== $* : exec->exit $(( (t1 - t0) / 1000000 )) ms"; grep "squid +" /tmp/cold_x.err | tail -3
}
run A=1
run A=1
run SQUID_IL_DEPTH=5
run SQUID_IL_DEPTH=4
run SQUID_TIDY_EXIT=1

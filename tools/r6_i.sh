#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6i /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for prio in 0 1 2; do for d in 3 5; do
  SQUID_TOK_PRIO=$prio SQUID_IL_DEPTH=$d timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 6 > gpurun_out/r6i/staged_p${prio}_d$d.log 2>&1
  echo "prio $prio depth $d: $(tail -1 gpurun_out/r6i/staged_p${prio}_d$d.log)"
done; done
SQUID_TOK_PRIO=1 tools/ingest_trace.sh /tmp/squid_bench/C3 p1 > gpurun_out/r6i/trace_p1.log 2>&1
head -12 gpurun_out/r6i/trace_p1.log | cut -c1-200

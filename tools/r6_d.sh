#!/bin/bash
# staged + from-file steps and the timeline with the current token pass
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6d /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for mode in 1 0; do
  SQUID_TOK_SPEC=$mode timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 7 > gpurun_out/r6d/staged_spec$mode.log 2>&1
  tail -1 gpurun_out/r6d/staged_spec$mode.log
done
for d in 2 3 4; do
  SQUID_IL_DEPTH=$d timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 6 > gpurun_out/r6d/staged_depth$d.log 2>&1
  echo "depth $d: $(tail -1 gpurun_out/r6d/staged_depth$d.log)"
done
SQUID_TOK_SPEC=1 timeout 600 python tools/file_step_timeline.py /tmp/squid_bench/C3 > gpurun_out/r6d/file_spec1.log 2>&1
grep "== " gpurun_out/r6d/file_spec1.log
SQUID_TOK_SPEC=1 tools/ingest_trace.sh /tmp/squid_bench/C3 spec1 > gpurun_out/r6d/trace_spec1.log 2>&1
head -24 gpurun_out/r6d/trace_spec1.log | cut -c1-400

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6e /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for cap in 1024 512 256; do for d in 2 3; do
  SQUID_TOK_CAP_MB=$cap SQUID_IL_DEPTH=$d timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 6 > gpurun_out/r6e/staged_cap${cap}_d$d.log 2>&1
  echo "cap $cap depth $d: $(tail -1 gpurun_out/r6e/staged_cap${cap}_d$d.log)"
done; done
for cap in 1024 512; do
  SQUID_TOK_CAP_MB=$cap SQUID_IL_DEPTH=3 timeout 600 python tools/file_step_timeline.py /tmp/squid_bench/C3 > gpurun_out/r6e/file_cap$cap.log 2>&1
  echo "file cap $cap: $(grep '== step' gpurun_out/r6e/file_cap$cap.log | tr '\n' ' ')"
done

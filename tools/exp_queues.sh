#!/bin/bash
# A/B of the ingest switches on staged C3 steps (GPU box)
build/gen_synth_bam --config C3 --out /tmp/c3 --threads 64 > /dev/null
run() { echo "-- $*"; env "$@" python3 tools/staged_steps.py /tmp/c3 7 2>&1 | grep "^== steps"; }
run SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 SQUID_IL_SPREAD=1
run SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 SQUID_IL_SPREAD=1 GPU_MAX_HW_QUEUES=8
run SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 GPU_MAX_HW_QUEUES=8
run SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 GPU_MAX_HW_QUEUES=12
run SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 GPU_MAX_HW_QUEUES=8 SQUID_TOK_CAP_MB=1536
echo "== from file"
for e in "SQUID_RESOLVE_GLOBAL=0" "SQUID_RESOLVE_GLOBAL=1 SQUID_IL_DEPTH=8 SQUID_TOK_WPB=1 GPU_MAX_HW_QUEUES=8"; do echo "-- $e"; env $e python3 tools/file_step_timeline.py /tmp/c3 2>&1 | grep "^== step"; done

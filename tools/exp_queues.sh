#!/bin/bash
# A/B of the ingest switches on C3 steps (GPU box)
build/gen_synth_bam --config C3 --out /tmp/c3 --threads 64 > /dev/null
run() { echo "-- $*"; env "$@" python3 tools/staged_steps.py /tmp/c3 7 2>&1 | grep "^== steps"; }
run SQUID_IL_DEPTH=8
run SQUID_IL_DEPTH=6
echo "== from file"
for e in "SQUID_IL_DEPTH=8" "SQUID_IL_DEPTH=6" "SQUID_IL_DEPTH=8 SQUID_TOK_RAMP_MB=256"; do echo "-- $e"; env $e python3 tools/file_step_timeline.py /tmp/c3 2>&1 | grep "^== step"; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6t2 gpurun_out/r6pp
PROBE_KB=0,18,22,28,40 timeout 900 python3 tools/parse_probe.py 1200000 2>&1 | grep -v "^GPU ingest\|^ingest" | tee gpurun_out/r6pp/probe2.txt | tail -16

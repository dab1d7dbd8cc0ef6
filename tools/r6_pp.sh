#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6pp
timeout 900 python3 tools/parse_probe.py 300000 2>&1 | grep -v "^GPU ingest\|^ingest" | tee gpurun_out/r6pp/probe.txt | tail -5

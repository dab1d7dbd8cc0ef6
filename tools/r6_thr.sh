#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6thr
python3 tools/throttle_probe.py C3 2>&1 | grep -v "^GPU ingest" | tee gpurun_out/r6thr/c3.txt | tail -8

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6fb
( time python bench.py ) > gpurun_out/r6fb/bench.json 2> gpurun_out/r6fb/bench.err
echo "bench rc $?"; grep "bench +" gpurun_out/r6fb/bench.err | cut -c1-150

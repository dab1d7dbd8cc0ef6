#!/bin/bash
# the whole GPU suite with the reader's new paths forced: two buffer sets, a carry room of 64 bytes (nearly every batch takes the buffer of its own)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 SQUID_IL_DEPTH=2 SQUID_CARRY_ROOM=64
mkdir -p gpurun_out/r6st
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6st/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6st/pytest_gpu.log
tail -4 gpurun_out/r6st/pytest_gpu.log | cut -c1-300

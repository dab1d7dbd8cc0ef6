#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6t2
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_matches or reader or inflate or token or route or damaged" > gpurun_out/r6t2/pytest.log 2>&1; tail -15 gpurun_out/r6t2/pytest.log | cut -c1-400

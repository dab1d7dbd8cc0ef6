#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6p
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "token_pass_variant or default_route" > gpurun_out/r6p/pytest.log 2>&1; tail -3 gpurun_out/r6p/pytest.log
( time python bench.py ) > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
echo "bench rc $?"; tail -2 gpurun_out/r6p/bench.err

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6lit
timeout 900 python -m pytest tests/test_literal_build_node.py -m gpu -x -q -k "gen3" > gpurun_out/r6lit/pytest.log 2>&1; tail -3 gpurun_out/r6lit/pytest.log | cut -c1-200

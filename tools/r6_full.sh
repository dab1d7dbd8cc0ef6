#!/bin/bash
# the whole GPU suite + smoke, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6f
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6f/pytest_gpu.log
tail -4 gpurun_out/r6f/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6f/smoke.log 2>&1; tail -1 gpurun_out/r6f/smoke.log
( time python bench.py ) > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err
echo "bench rc $?"; tail -4 gpurun_out/r6f/bench.err | cut -c1-300

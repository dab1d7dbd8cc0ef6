#!/bin/bash
# the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6s
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6s/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6s/pytest_gpu.log
tail -8 gpurun_out/r6s/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6s/smoke.log 2>&1; tail -2 gpurun_out/r6s/smoke.log

#!/bin/bash
# the whole GPU suite + smoke, then a short bench line without the side legs (timing of a change)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6q
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6q/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6q/pytest_gpu.log
tail -4 gpurun_out/r6q/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6q/smoke.log 2>&1; tail -1 gpurun_out/r6q/smoke.log
( time python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli ) > gpurun_out/r6q/bench.json 2> gpurun_out/r6q/bench.err
echo "bench rc $?"; tail -3 gpurun_out/r6q/bench.err

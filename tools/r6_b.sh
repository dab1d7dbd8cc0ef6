#!/bin/bash
# a short bench line without the side legs (timing of a change)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6b
( time python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli ) > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err
echo "bench rc $?"; tail -3 gpurun_out/r6b/bench.err

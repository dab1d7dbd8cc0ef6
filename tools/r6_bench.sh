#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6bench
( time python bench.py ) > gpurun_out/r6bench/bench.json 2> gpurun_out/r6bench/bench.err
echo "rc $?"; tail -3 gpurun_out/r6bench/bench.err; tail -c 600 gpurun_out/r6bench/bench.json

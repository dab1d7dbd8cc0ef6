#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6l /tmp/squid_bench
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or reader or four_million" > gpurun_out/r6l/pytest_reader.log 2>&1; tail -2 gpurun_out/r6l/pytest_reader.log
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
timeout 1200 python tools/shard_project.py /tmp/squid_bench/C3 2 4 8 > gpurun_out/r6l/shard_projection_C3.log 2>&1
cat gpurun_out/r6l/shard_projection_C3.log | cut -c1-400

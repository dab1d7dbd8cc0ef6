#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6j /tmp/squid_bench
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or reader or damaged or both_files or default_route" > gpurun_out/r6j/pytest_reader.log 2>&1; tail -2 gpurun_out/r6j/pytest_reader.log
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 7 > gpurun_out/r6j/staged.log 2>&1; tail -1 gpurun_out/r6j/staged.log
tools/ingest_trace.sh /tmp/squid_bench/C3 cur > gpurun_out/r6j/trace.log 2>&1
head -16 gpurun_out/r6j/trace.log | cut -c1-200

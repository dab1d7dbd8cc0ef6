#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6n /tmp/squid_bench
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or reader or four_million" > gpurun_out/r6n/pytest_reader.log 2>&1; tail -2 gpurun_out/r6n/pytest_reader.log
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 7 > gpurun_out/r6n/staged.log 2>&1; tail -1 gpurun_out/r6n/staged.log
for i in 1 2; do timeout 600 python tools/file_step_timeline.py /tmp/squid_bench/C3 > gpurun_out/r6n/file$i.log 2>&1; echo "$(grep '== step' gpurun_out/r6n/file$i.log | sed 's/== step [0-9]: //' | tr '\n' '|')"; grep "file pieces queued" gpurun_out/r6n/file$i.log | tail -3 | cut -c1-140; done

#!/bin/bash
# round 6, first GPU run of the speculative token pass: the GPU-reader tests, then staged C3 steps with the new and the old pass
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or reader or damaged or both_files" > gpurun_out/r6a/pytest_reader.log 2>&1
echo "pytest rc $?" >> gpurun_out/r6a/pytest_reader.log
tail -5 gpurun_out/r6a/pytest_reader.log
mkdir -p /tmp/squid_bench
( time build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 ) > gpurun_out/r6a/synth.log 2>&1
ls -la /tmp/squid_bench >> gpurun_out/r6a/synth.log
SQUID_INFLATE_CHECK=1 timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 1 > gpurun_out/r6a/check_spec.log 2>&1
grep -c "inflate check" gpurun_out/r6a/check_spec.log; grep "inflate check" gpurun_out/r6a/check_spec.log | tail -2
for mode in 1 0 1; do
  SQUID_TOK_SPEC=$mode timeout 600 python tools/staged_steps.py /tmp/squid_bench/C3 7 > gpurun_out/r6a/staged_spec$mode.log 2>&1
  tail -1 gpurun_out/r6a/staged_spec$mode.log
done
SQUID_TOK_SPEC=1 timeout 600 python tools/file_step_timeline.py /tmp/squid_bench/C3 > gpurun_out/r6a/file_spec1.log 2>&1
grep "== " gpurun_out/r6a/file_spec1.log

#!/bin/bash
# every BGZF block of the C3 files through the final reader against zlib (SQUID_INFLATE_CHECK), then the stress settings on the GPU suite's reader tests
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p /tmp/squid_bench gpurun_out/r6chk
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
SQUID_INFLATE_CHECK=1 timeout 900 python tools/staged_steps.py /tmp/squid_bench/C3 1 > gpurun_out/r6chk/check.log 2>&1
grep "inflate check" gpurun_out/r6chk/check.log | tail -2 | cut -c1-200; grep -c "inflate check" gpurun_out/r6chk/check.log
SQUID_IL_DEPTH=2 SQUID_CARRY_ROOM=64 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_matches or reader or inflate or token or route or damaged or four_million" > gpurun_out/r6chk/pytest.log 2>&1; tail -2 gpurun_out/r6chk/pytest.log

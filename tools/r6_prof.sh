#!/bin/bash
# round 6 profiles: rocprofv3 --kernel-trace --stats of the bench command; SQ / LDS / HBM counters of the ingest kernels over staged steps
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
mkdir -p /tmp/squid_bench gpurun_out
tools/profile.sh r06 --no-cold-cli > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log | cut -c1-300
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
tools/pmc_ingest.sh /tmp/squid_bench/C3 r06 > gpurun_out/r06_pmc_ingest.txt 2>&1; cat gpurun_out/r06_pmc_ingest.txt

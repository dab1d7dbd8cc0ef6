#!/bin/bash
# round 6: the inflate kernels alone (tools/tok_bench.py) + the timeline of a staged step with either token pass
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6b /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 51211 51210 25610 25611 38411 102411 2 > gpurun_out/r6b/tok_bench.log 2>&1
cat gpurun_out/r6b/tok_bench.log
python tools/tok_bench.py /tmp/squid_bench/C3.bam 2048 3 51211 2 >> gpurun_out/r6b/tok_bench.log 2>&1
tail -2 gpurun_out/r6b/tok_bench.log
SQUID_TOK_SPEC=1 tools/ingest_trace.sh /tmp/squid_bench/C3 spec1 > gpurun_out/r6b/trace_spec1.log 2>&1
SQUID_TOK_SPEC=0 tools/ingest_trace.sh /tmp/squid_bench/C3 spec0 > gpurun_out/r6b/trace_spec0.log 2>&1
cat gpurun_out/r6b/trace_spec1.log

#!/bin/bash
# the ordering solver of round 6 on the GPU box: solver tests, --bwa tests, the 1 M-record --bwa sample against the oracle, and the full-size --bwa sample
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6g /tmp/squid_bench
timeout 1500 python -m pytest tests -m gpu -x -q -k "above_26 or bwa or order or giant" > gpurun_out/r6g/pytest.log 2>&1; tail -3 gpurun_out/r6g/pytest.log
build/gen_synth_bam --config C3 --bwa --records 1000000 --seed 20180003 --out /tmp/squid_bench/c3bwa1m > /dev/null 2>&1
ORACLE_STATS_FILE=gpurun_out/r6g/oracle_stats_1m.txt build/squid_oracle --bwa -b /tmp/squid_bench/c3bwa1m.bam -o /tmp/squid_bench/o1m > /dev/null 2> gpurun_out/r6g/oracle_1m.err
build/squid --bwa -b /tmp/squid_bench/c3bwa1m.bam -o /tmp/squid_bench/g1m > /dev/null 2> gpurun_out/r6g/squid_1m.err
cmp /tmp/squid_bench/o1m_sv.txt /tmp/squid_bench/g1m_sv.txt && echo "1M --bwa sample: _sv.txt identical, $(wc -l < /tmp/squid_bench/g1m_sv.txt) lines"; cat gpurun_out/r6g/oracle_stats_1m.txt | tr '\n' ' '; cat gpurun_out/r6g/squid_1m.err | tail -3
python tools/bwa_probe.py --steps 2 > gpurun_out/r6g/bwa_full.json 2> gpurun_out/r6g/bwa_full.err; tail -4 gpurun_out/r6g/bwa_full.err; cut -c1-600 gpurun_out/r6g/bwa_full.json

#!/bin/bash
# SQ counters of the ingest kernels (first two counter sets of tools/pmc_ingest.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
mkdir -p /tmp/squid_bench gpurun_out
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
sed 's/ "FETCH_SIZE" "WRITE_SIZE"; do/; do/' tools/pmc_ingest.sh > /tmp/pmc2.sh; chmod +x /tmp/pmc2.sh; cp /tmp/pmc2.sh tools/.pmc2_tmp.sh
tools/.pmc2_tmp.sh /tmp/squid_bench/C3 r06b > gpurun_out/r06b_pmc_ingest.txt 2>&1; rm -f tools/.pmc2_tmp.sh; cat gpurun_out/r06b_pmc_ingest.txt

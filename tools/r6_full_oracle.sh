#!/bin/bash
# round 6, VERDICT item 2a: the CPU oracle on the FULL C5 and C4 samples (one core, minutes each), its _sv.txt against the library's
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6o
( time python bench.py --workload C5 --steps 2 --warmup 1 --no-cold-cli --resident-steps 1 ) > gpurun_out/r6o/dense_full.json 2> gpurun_out/r6o/dense_full.err
echo "C5 rc $?"; tail -c 1500 gpurun_out/r6o/dense_full.json | head -c 1500; echo
rm -f /tmp/squid_bench/C5_*
( time python bench.py --workload C4 --steps 2 --warmup 1 --no-cold-cli --resident-steps 1 ) > gpurun_out/r6o/c4_full.json 2> gpurun_out/r6o/c4_full.err
echo "C4 rc $?"; tail -c 1500 gpurun_out/r6o/c4_full.json

#!/bin/bash
# HBM traffic per kernel from the TCC counters (run on the GPU box through gpurun).  Two separate passes, counters
# only (no --stats / trace domains), as MI355X_MICROARCH.md prescribes; a 1 GiB calibration read (k_calib_read4) in
# the same run gives the byte-per-count factor for this access shape.  Usage: tools/profile_pmc.sh <tag>
set -u
TAG=${1:-r01}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp SQUID_CALIB=1
mkdir -p gpurun_out/pmc_$TAG
for CNT in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$TAG/$CNT -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc_$TAG/$CNT.log 2>&1
  find gpurun_out/pmc_$TAG/$CNT -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_${TAG}_$CNT.csv
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE.csv gpurun_out/pmc_${TAG}_WRITE_SIZE.csv > gpurun_out/pmc_${TAG}_traffic.json
cat gpurun_out/pmc_${TAG}_traffic.json | head -50

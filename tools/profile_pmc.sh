#!/bin/bash
# Counters per kernel of the graph pass over the resident C3 records (run on the GPU box through gpurun).  Separate passes,
# counters only (no --stats / no trace domains besides the kernel trace), as MI355X_MICROARCH.md prescribes:
#   FETCH_SIZE, WRITE_SIZE  -> HBM traffic per launch; a 1 GiB calibration read (k_calib_read4, SQUID_CALIB=1) in the same run
#                              gives the bytes-per-count factor for the access shape of the record scans
#   SQ_*                    -> where the waves' cycles go
# Usage: tools/profile_pmc.sh <tag>
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp SQUID_CALIB=1
mkdir -p /tmp/c3 gpurun_out/pmc_$TAG
[ -f /tmp/c3/C3.bam ] || build/gen_synth_bam --config C3 --out /tmp/c3/C3 --threads 64 > /dev/null
for CNT in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
  NAME=$(echo $CNT | cut -d' ' -f1)
  ( cd /tmp && rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$TAG/$NAME -o pmc -- python3 $REPO/tools/resident_pass.py /tmp/c3/C3 3 ) > gpurun_out/pmc_$TAG/$NAME.log 2>&1
  find gpurun_out/pmc_$TAG/$NAME -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_${TAG}_$NAME.csv
  rm -rf gpurun_out/pmc_$TAG/$NAME
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE.csv gpurun_out/pmc_${TAG}_WRITE_SIZE.csv C3 > gpurun_out/pmc_${TAG}_traffic.json
python3 tools/pmc_sq_summary.py gpurun_out/pmc_${TAG}_SQ_WAVE_CYCLES.csv > gpurun_out/pmc_${TAG}_sq.txt
head -60 gpurun_out/pmc_${TAG}_traffic.json
cat gpurun_out/pmc_${TAG}_sq.txt

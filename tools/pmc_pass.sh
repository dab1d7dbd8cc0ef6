#!/bin/bash
# SQ counters of the graph-pass kernels over resident records (quick loop): tools/pmc_pass.sh <tag> [records]
set -u
TAG=${1:-x}; REC=${2:-20000000}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
python3 tools/pass_timing.py --records $REC --passes 1 > /dev/null 2>&1   # generates the sample once
for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  NAME=$(echo $CNT | cut -d' ' -f1)
  ( cd /tmp && timeout 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$TAG/$NAME -o pmc -- python3 $REPO/tools/pass_timing.py --records $REC --passes 2 ) > gpurun_out/pmc_$TAG/$NAME.log 2>&1
  find gpurun_out/pmc_$TAG/$NAME -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_${TAG}_$NAME.csv
  rm -rf gpurun_out/pmc_$TAG/$NAME
done
python3 tools/pmc_sq_summary.py gpurun_out/pmc_${TAG}_SQ_WAVE_CYCLES.csv | head -14
python3 - <<PY
import csv
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open("gpurun_out/pmc_${TAG}_SQ_INSTS_LDS.csv")):
    k=(r.get("Kernel_Name") or "").split("(")[0].replace("void ","").replace("sq::","")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:8]:
    w=c.get("SQ_WAVES",1) or 1
    print(f"{k[:36]:36s} lds/wave {c.get('SQ_INSTS_LDS',0)/w:8.1f} vmem_wr/wave {c.get('SQ_INSTS_VMEM_WR',0)/w:7.1f} smem/wave {c.get('SQ_INSTS_SMEM',0)/w:7.1f} wait_lds {c.get('SQ_WAIT_INST_LDS',0)/max(c.get('SQ_WAVE_CYCLES',1),1):5.2f} waves {w:.0f}")
PY

#!/bin/bash
# SQ / LDS counters of the ingest kernels over staged C3 steps (rocprofv3 --pmc, counters only + kernel trace; separate passes).  GPU box.
# Usage: tools/pmc_ingest.sh <prefix of the C3 BAM pair> <tag>
set -u
PRE=$1; TAG=${2:-r04}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmci_$TAG
i=0
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $REPO/gpurun_out/pmci_$TAG/p$i -o pmc -- python3 $REPO/tools/staged_steps.py $PRE 3 ) > gpurun_out/pmci_$TAG/p$i.log 2>&1
  F=$(find gpurun_out/pmci_$TAG/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "").replace("sq::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
for k in ("k_inflate_spec<384, 10>", "k_inflate_spec<256, 10>", "k_inflate_tok2<false>", "k_lz_resolve5", "k_lz_resolve3", "k_parse_records", "k_parse_place", "k_rec_sync"):
    if k in acc:
        c = acc[k]; w = c.get("SQ_WAVES", 0) or 1
        print(f"{k:24s} launches {len(n[k]):3d} waves {w:9.0f} | " + " ".join(f"{name[3:]}/wave {v / w:.4g}" for name, v in sorted(c.items()) if name != "SQ_WAVES"))
PY
  rm -rf gpurun_out/pmci_$TAG/p$i
done

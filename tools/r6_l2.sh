#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r6l2
for back in 400 2000; do build/l2_probe 8192 300 $back; done | tee gpurun_out/r6l2/times.txt
rm -rf /tmp/r6l2
( cd /tmp && timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d /tmp/r6l2 -o pmc -- $REPO/build/l2_probe 8192 300 400 ) > gpurun_out/r6l2/pmc.log 2>&1
F=$(find /tmp/r6l2 -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY' | tee gpurun_out/r6l2/counters.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r.get("Kernel_Name") or "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
for k in sorted(acc):
    print(f"{k:40s} launches {len(n[k])} | " + " ".join(f"{name} {v / len(n[k]):.4g}" for name, v in sorted(acc[k].items())))
PY

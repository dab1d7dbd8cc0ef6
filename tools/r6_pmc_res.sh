#!/bin/bash
# L2 counters of the resolve alone (tools/tok_bench.py: 16384 blocks of the C3 file, every kernel by itself)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
REPO=$PWD
mkdir -p /tmp/squid_bench gpurun_out/r6pr
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
i=0
for CNT in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i + 1))
  rm -rf /tmp/r6pr$i
  ( cd /tmp && timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/r6pr$i -o pmc -- python3 $REPO/tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 2 25610 ) > gpurun_out/r6pr/p$i.log 2>&1
  F=$(find /tmp/r6pr$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); sys.exit(0)
for r in rows:
    k = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "").replace("sq::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
for k in acc:
    if k.startswith("k_lz_resolve") or k.startswith("k_inflate_spec"):
        print(f"{k:26s} launches {len(n[k]):3d} | " + " ".join(f"{name} {v / len(n[k]):.4g}" for name, v in sorted(acc[k].items())))
PY
done
tail -3 gpurun_out/r6pr/p1.log | cut -c1-300

#!/usr/bin/env python3
"""per-kernel SQ counters from one rocprofv3 --pmc pass: share of the waves' cycles spent parked (WAIT_ANY), issue-stalled
(WAIT_INST_ANY) and issuing (ACTIVE_INST_ANY), instructions per wave"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "").replace("sq::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r.get("Dispatch_Id"))
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]
print(f"{'kernel':40s} {'launches':>8s} {'wave_cyc/launch':>16s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'valu/wave':>10s} {'vmem_rd/wave':>13s} {'salu/wave':>10s}")
for k, c in rows:
    n = max(1, len(calls[k])); wc = c.get("SQ_WAVE_CYCLES", 0) or 1; w = c.get("SQ_WAVES", 0) or 1
    print(f"{k[:40]:40s} {n:8d} {wc / n:16.3e} {c.get('SQ_WAIT_ANY', 0) / wc:7.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:8.2f} {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.2f} {c.get('SQ_INSTS_VALU', 0) / w:10.1f} {c.get('SQ_INSTS_VMEM_RD', 0) / w:13.1f} {c.get('SQ_INSTS_SALU', 0) / w:10.1f}")

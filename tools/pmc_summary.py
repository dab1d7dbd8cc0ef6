#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Counter units are calibrated on k_calib_read4 (a coalesced 4-byte-per-lane read of exactly 2^30 bytes in the same
run): bytes_per_count = 2^30 / FETCH_SIZE(k_calib_read4) (2048 on gfx950: KiB, and the guide's factor of two for coalesced
reads).  WRITE_SIZE does not carry the factor of two: the scan that writes exactly 4 B per record (203.3 MB at C3) shows
200 690 counts, i.e. 1024 B per count -- so writes are scaled by half the read factor."""
import csv
import json
import re
import sys
from collections import defaultdict

# kernels that one HIP-event timer of the library brackets together (bench.py looks traffic up by timer name)
GROUPS = {
    "k_depth_check+fix": ["k_depth_check", "k_depth2<true>", "k_fold_stripes"],
    "k_bp_key_prefix": ["k_bp_key_reduce", "k_bp_key_scan"],
    "k_bp_walk": ["k_bp_walk2<false>", "k_bp_chain", "k_bp_walk2<true>"],
    "k_tile_scan": ["k_tile_partial", "k_tile_scan"],
}


def short(name):
    n = name.split("(")[0].replace("sq::", "").replace("void ", "").strip()
    return n if n in sum(GROUPS.values(), []) else re.sub(r"<.*>$", "", n)  # (k_pass1<false, 5> -> k_pass1)  # k_pass1<false> -> k_pass1; the grouped ones keep their template argument


def load(path):
    per = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            per[short(name)].append(val)
    return per


fetch, write = load(sys.argv[1]), load(sys.argv[2])
cal = [v for k, vs in fetch.items() if "k_calib_read4" in k for v in vs]
factor = (1 << 30) / (sum(cal) / len(cal)) if cal else None
out = {"_workload": sys.argv[3] if len(sys.argv) > 3 else None,
       "_calibration": {"kernel": "k_calib_read4", "bytes": 1 << 30, "fetch_counts": cal, "bytes_per_count": factor}}
for k in sorted(fetch):
    if "k_calib" in k or not (k.startswith("k_") or "k_scan" in k):
        continue
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
    out[k] = {"launches": len(fetch[k]), "fetch_bytes_per_launch": f * factor if factor else None, "write_bytes_per_launch": w * factor / 2 if factor else None,
              "hbm_bytes_per_launch": (f + w / 2) * factor if factor else None}
for g, members in GROUPS.items():  # per launch of the group = one launch of each member
    if all(m in out for m in members):
        out[g] = {"launches": min(out[m]["launches"] for m in members), "members": members,
                  **{f: sum(out[m][f] for m in members) for f in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}}
print(json.dumps(out, indent=1))

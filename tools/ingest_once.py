#!/usr/bin/env python3
"""one ingest of a BAM file (for rocprofv3: tools/profile_ingest.sh).  usage: ingest_once.py <prefix>"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre = sys.argv[1]
with squid_amd.Context() as ctx:
    t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16); dt = time.time() - t0
    n = ctx.counts()["n_concordant"]
    print(f"{n} records in {dt*1e3:.0f} ms ({n/dt/1e6:.1f} M rec/s)")
    for k, v in ctx.timing().items():
        if any(s in k for s in ("infl", "lz_", "rec_", "parse")): print("  ", k, round(v["ms"], 1), "ms,", v["launches"], "launches")

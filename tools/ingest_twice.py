#!/usr/bin/env python3
"""two ingests of a BAM file staged in HBM (the second one is the warm one the timeline tools look at).  usage: ingest_twice.py <prefix>"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre = sys.argv[1]; nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
with squid_amd.Context() as ctx:
    ctx.stage_bam(f"{pre}.bam")
    for it in range(2):
        ctx.clear_records()
        t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=nthreads); dt = time.time() - t0
        n = ctx.counts()["n_concordant"]
        print(f"ingest {it}: {n} records in {dt*1e3:.0f} ms ({n/dt/1e6:.1f} M rec/s)")

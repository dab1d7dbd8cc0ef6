#!/usr/bin/env python3
"""three staged steps of the bench workload (compressed BAM resident in HBM), for a profiler to wrap"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre = sys.argv[1]
with squid_amd.Context() as ctx:
    ctx.stage_bam(f"{pre}.bam")
    for it in range(3):
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=256)
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); ctx.sv_text_fast()
        print(f"== staged step {it}: load {1e3 * (t1 - t0):.1f} ms, whole step {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)

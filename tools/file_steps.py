#!/usr/bin/env python3
"""N from-file steps of the bench workload (for a profiler): file_steps.py <prefix> <N>"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre = sys.argv[1] if len(sys.argv) > 1 else "/tmp/squid_bench/C3_s20180003"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with squid_amd.Context() as ctx:
    for it in range(n):
        squid_amd.drop_file_cache()
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=256)
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); ctx.sv_text_fast()
        print(f"== step {it}: load {1e3 * (t1 - t0):.1f} ms, whole step {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
        time.sleep(0.3)

#!/usr/bin/env python3
"""staged steps of the bench workload (compressed BAM resident in HBM): N steps, whole-step wall times; for A/B runs of environment
switches and for a profiler to wrap.  Usage: staged_steps.py <prefix> [N]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ts = []
with squid_amd.Context() as ctx:
    ctx.stage_bam(f"{pre}.bam")
    for it in range(n):
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=256)
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); sv = ctx.sv_text_fast()
        ts.append(1e3 * (time.perf_counter() - t0))
        print(f"== staged step {it}: load {1e3 * (t1 - t0):.1f} ms, whole step {ts[-1]:.1f} ms", flush=True)
import hashlib
print("== steps", " ".join(f"{t:.0f}" for t in ts), "| median of the last", n - 2, f"{sorted(ts[2:])[len(ts[2:]) // 2]:.1f} ms | sv sha {hashlib.sha256(sv if isinstance(sv, bytes) else sv.encode()).hexdigest()[:12]}", flush=True)

#!/usr/bin/env python3
"""from-file steps of the bench workload with the reader's timeline on stderr (SQUID_INGEST_TIMING)"""
import os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.init(); torch.zeros(8, device="cuda:0"); torch.cuda.synchronize()
import squid_amd
pre = sys.argv[1] if len(sys.argv) > 1 else "/tmp/squid_bench/C3_s20180003"
os.environ["SQUID_INGEST_TIMING"] = "1"
with squid_amd.Context() as ctx:
    for it in range(4):
        squid_amd.drop_file_cache()
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=int(os.environ.get("THREADS", "256")))
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); ctx.sv_text_fast()
        print(f"== step {it}: load {1e3 * (t1 - t0):.1f} ms, whole step {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    ctx.stage_bam(f"{pre}.bam")
    for it in range(3):
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=int(os.environ.get("THREADS", "256")))
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); ctx.sv_text_fast()
        print(f"== staged step {it}: load {1e3 * (t1 - t0):.1f} ms, whole step {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)

#!/usr/bin/env python3
"""per-kernel table of the graph pass over resident records: ms, launches, algorithmic bytes and GB/s per pass, by time.
usage: kernel_table.py <prefix> [passes]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
with squid_amd.Context() as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
    ctx.reset(); ctx.build_graph(); ctx.order(); ctx.sv_text()  # warm
    ctx.timing_accumulate(True)
    ctx.timing()  # (clears)
    for _ in range(n):
        ctx.reset(); ctx.build_graph(); ctx.order(); ctx.sv_text()
    t = ctx.timing()
    tot_ms = tot_b = 0.0
    for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms"]):
        ms, b = v["ms"] / n, v["bytes"] / n
        if k.startswith(("k_", "scan_")) and b > 0:
            tot_ms += ms; tot_b += b
        print(f"{k:28s} {ms:8.3f} ms  x{v['launches'] / n:5.1f}  {b / 1e9:7.3f} GB  {b / max(ms, 1e-9) / 1e6:8.1f} GB/s")
    print(f"record-streaming kernels: {tot_ms:.2f} ms, {tot_b / 1e9:.2f} GB per pass")

#!/usr/bin/env python3
"""one ingest of the C3 workload, then N graph passes over the resident records (for rocprofv3 --pmc: tools/profile_pmc.sh).
usage: resident_pass.py <prefix> [passes]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd
pre, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
with squid_amd.Context() as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
    for _ in range(n):
        ctx.reset(); ctx.build_graph(); ctx.order(); text = ctx.sv_text()
    print(ctx.counts()["n_concordant"], "records,", text.count("\n") - 1, "SV rows")

#!/usr/bin/env python3
"""k_edges with parts switched off (SQUID_EDGES_ABLATE bits: 1 no edge emission, 2 stop behind the block-0 fit, 4 stop behind the
first record load): where its time goes.  Run per setting in a fresh process: usage edges_ablate.py <prefix>"""
import os, subprocess, sys
pre = sys.argv[1]
code = ("import sys; sys.path.insert(0, %r); import squid_amd\n"
        "ctx = squid_amd.Context(); ctx.load(%r, %r, threads=16)\n"
        "ctx.timing_accumulate(True)\n"
        "for _ in range(3):\n"
        "    ctx.reset()\n"
        "    try: ctx.build_graph()\n"
        "    except Exception as e: pass\n"
        "t = ctx.timing(); print({k: round(v['ms'] / max(1, v['launches']), 3) for k, v in t.items() if k in ('k_edges', 'k_edges_near', 'k_dedup', 'k_classify', 'k_depth')}, ctx.counts()['n_raw_edges'], ctx.counts()['n_unique_edges'])\n") % (
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pre + ".bam", pre + ".chim.bam")
for ab in ("0", "1", "2", "4"):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SQUID_EDGES_ABLATE=ab), capture_output=True, text=True)
    print("ablate", ab, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])

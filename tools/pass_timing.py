#!/usr/bin/env python3
"""per-kernel HIP-event timings of the graph pass over resident records (quick loop while tuning the record kernels).
usage: pass_timing.py [--config C3] [--records N] [--passes K]"""
import argparse, os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--records", type=int, default=10_000_000); ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--tsv", type=int, default=0)
a = ap.parse_args()
pre = Path(tempfile.gettempdir()) / f"pt_{a.config}_{a.records}_{a.tsv}"
if not Path(f"{pre}.bam").exists():
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", a.config, "--records", str(a.records), "--out", str(pre), "--threads", str(os.cpu_count() or 8)] + (["--tsv", str(a.tsv)] if a.tsv else []), stdout=subprocess.DEVNULL)
kw = {}
if a.config.startswith("C5"):
    kw = dict(min_edge_weight=1, max_allowed_degree=50)
with squid_amd.Context(**kw) as ctx:
    t0 = time.perf_counter()
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
    print(f"load {time.perf_counter() - t0:.3f} s")
    ctx.reset(); ctx.build_graph(); ctx.order(); ctx.sv_text()
    ctx.timing_accumulate(True)
    t0 = time.perf_counter()
    for _ in range(a.passes):
        ctx.reset(); ctx.build_graph(); ctx.order(); text = ctx.sv_text()
    dt = (time.perf_counter() - t0) / a.passes
    n = ctx.counts()["n_concordant"]
    print(f"{n} records, {text.count(chr(10)) - 1} SV rows, {dt * 1e3:.2f} ms per pass, counts {ctx.counts()}")
    for k, v in sorted(ctx.timing().items(), key=lambda kv: -kv[1]["ms"]):
        ms = v["ms"] / a.passes
        gbs = v["bytes"] / max(v["ms"], 1e-9) / 1e6
        print(f"  {k:28s} {ms:9.4f} ms  x{v['launches'] // a.passes:<4d} {gbs:9.1f} GB/s")

#!/usr/bin/env python3
"""--bwa mode on the C3 sample: where does the time go?  (sq_ingest_bwa_file, sq_build_graph, ordering, SV calls)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd

pre = sys.argv[1] if len(sys.argv) > 1 else "/tmp/squid_bench/C3_s20180003"
if not os.path.exists(pre + ".bam"):
    import subprocess
    os.makedirs(os.path.dirname(pre), exist_ok=True)
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", "C3", "--seed", pre.rsplit("_s", 1)[1], "--out", pre, "--threads", "128"], stdout=subprocess.DEVNULL)
squid_amd.keep_host_memory()
with squid_amd.Context(star_mapq=False) as ctx:
    ctx.keep_stage_graphs(False)
    for it in range(3):
        ctx.clear_records()
        t0 = time.perf_counter()
        ctx.load_bwa(f"{pre}.bam", threads=16)
        t1 = time.perf_counter()
        ctx.build_graph()
        t2 = time.perf_counter()
        ctx.order_sizes()
        text = ctx.sv_text_fast()
        t3 = time.perf_counter()
        k = ctx.counts()
        import hashlib
        print(f"step {it}: ingest {1e3*(t1-t0):.0f} ms, graph {1e3*(t2-t1):.0f} ms, order + SV {1e3*(t3-t2):.0f} ms; {k['n_concordant']} records, {len(text.splitlines())} rows, sha {hashlib.sha256(text.encode()).hexdigest()[:12]}, "
              f"{k['n_concordant']/(t3-t0)/1e6:.1f} M aln/s", flush=True)
    if os.environ.get("SQUID_BWA_STAGES"):
        for a, b in sorted(ctx.timing().items(), key=lambda t: -t[1]["ms"])[:25]:
            print(f"   {a:32s} {b['ms']:9.1f} ms")

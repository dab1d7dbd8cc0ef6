#!/usr/bin/env python3
"""ingest timing probe (GPU box): phases of sq_ingest_concordant_file for a generator config"""
import os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
extra = sys.argv[2:]
os.environ["SQUID_INGEST_TIMING"] = "1"
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32", *extra], stdout=subprocess.DEVNULL)
    for threads in (16, 32, 48, 64, 96):
        with squid_amd.Context() as ctx:
            t0 = time.time(); names, lens = squid_amd.read_header(f"{pre}.bam"); t_h = time.time() - t0; t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=threads); dt = time.time() - t0
            print(f"threads={threads}: header {t_h*1e3:.1f} ms load {dt*1e3:.1f} ms, {ctx.counts()['n_concordant']/dt/1e6:.2f} M rec/s", flush=True)

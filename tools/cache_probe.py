#!/usr/bin/env python3
"""GPU box: BAM ingest against the record cache (sq_save_records / sq_load_records) on one generator config"""
import hashlib, os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
    cache = Path(td) / "records.sqsoa"
    with squid_amd.Context() as ctx:
        t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16); t_bam = time.time() - t0
        want = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in ctx.records().items()}
        t0 = time.time(); ctx.save_records(cache); t_save = time.time() - t0
        n = ctx.counts()["n_concordant"]
    for _ in range(2):
        with squid_amd.Context() as ctx:
            t0 = time.time(); ctx.load_cached(f"{pre}.bam", f"{pre}.chim.bam", cache); t_load = time.time() - t0
            same = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in ctx.records().items()} == want
        print(f"{cfg}: {n} records; from the BAM file {t_bam*1e3:.0f} ms; cache of {cache.stat().st_size/1e9:.2f} GB written in {t_save*1e3:.0f} ms, loaded in {t_load*1e3:.0f} ms ({n/t_load/1e6:.0f} M rec/s), identical arrays: {same}", flush=True)

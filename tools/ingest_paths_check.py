#!/usr/bin/env python3
"""GPU box: the resident record arrays after the host ingest pipeline and after the GPU one (SQUID_GPU_INFLATE=1) must be
the same bytes.  Usage: ingest_paths_check.py [config]   (default C3: 50.8 M records, 11 batches with carried records)"""
import hashlib, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
    got = {}
    for mode in ("0", "1"):
        os.environ["SQUID_GPU_INFLATE"] = mode
        with squid_amd.Context() as ctx:
            ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
            got[mode] = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in ctx.records().items()}
            print("SQUID_GPU_INFLATE=" + mode, ctx.counts()["n_concordant"], "records", flush=True)
    same = got["0"] == got["1"]
    print("identical arrays:", same, sorted(got["0"]))
    sys.exit(0 if same else 1)

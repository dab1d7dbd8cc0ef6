#!/usr/bin/env python3
"""GPU box: for several generator seeds and batch sizes, the record arrays after the host reader and after the GPU reader must
be the same bytes.  usage: reader_stress.py [records] [seeds]"""
import hashlib, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
records = sys.argv[1] if len(sys.argv) > 1 else "3000000"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = 0
with tempfile.TemporaryDirectory() as td:
    for seed in range(1, seeds + 1):
        pre = Path(td) / f"s{seed}"
        subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", "C3", "--records", records, "--seed", str(7000 + seed), "--indel-frac", "0.1", "--out", str(pre), "--threads", "32"], stdout=subprocess.DEVNULL)
        got = {}
        for mode, cap in (("0", None), ("1", "64"), ("1", "200"), ("1", None)):
            os.environ["SQUID_GPU_INFLATE"] = mode
            if cap: os.environ["SQUID_TOK_CAP_MB"] = cap
            else: os.environ.pop("SQUID_TOK_CAP_MB", None)
            with squid_amd.Context() as ctx:
                ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
                got[(mode, cap)] = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in ctx.records().items()}
        ref = got[("0", None)]
        ok = all(v == ref for v in got.values())
        bad += not ok
        print("seed", seed, "identical" if ok else "DIFFERENT", flush=True)
        for f in Path(td).glob(f"s{seed}.*"): f.unlink()
sys.exit(1 if bad else 0)

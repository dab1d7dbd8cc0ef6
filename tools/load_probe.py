#!/usr/bin/env python3
"""where Context.load() spends its time (GPU box)"""
import ctypes as C, os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
extra = sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    pre = Path(td) / cfg
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--out", str(pre), "--threads", "32", *extra], stdout=subprocess.DEVNULL)
    for rep in range(3):
        t = [time.perf_counter()]
        ctx = squid_amd.Context(); t.append(time.perf_counter())
        names, lens = squid_amd.read_header(f"{pre}.bam"); t.append(time.perf_counter())
        arr = (C.c_int32 * len(lens))(*lens)
        ctx.lib.sq_set_references(ctx.h, len(lens), arr); t.append(time.perf_counter())
        ctx.lib.sq_ingest_chimeric_file(ctx.h, f"{pre}.chim.bam".encode()); t.append(time.perf_counter())
        ctx.lib.sq_ingest_concordant_file(ctx.h, f"{pre}.bam".encode(), 32); t.append(time.perf_counter())
        ctx.close(); t.append(time.perf_counter())
        lab = ["create", "header", "set_refs", "chimeric", "concordant", "close"]
        print("  ".join(f"{l} {1e3 * (b - a):.1f}" for l, a, b in zip(lab, t, t[1:])), "ms", flush=True)

#!/usr/bin/env python3
"""repeat the GPU reader on a small file under several switch sets, in fresh processes: every run must take the GPU path (rc 0) and give the same record arrays"""
import hashlib, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
pre = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
code = ("import sys, json, hashlib; sys.path.insert(0, %r); import squid_amd\n"
        "ctx = squid_amd.Context(); ctx.load(sys.argv[1], sys.argv[2]); r = ctx.records()\n"
        "print(json.dumps({k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in r.items()}))") % str(ROOT)
def run(env):
    p = subprocess.run([sys.executable, "-c", code, pre + ".bam", pre + ".chim.bam"], env=dict(os.environ, **env), capture_output=True, text=True)
    ok = p.returncode == 0 and (not env or ("GPU inflate+parse path" in p.stderr and "(rc 0)" in p.stderr))
    return (p.stdout.strip().splitlines() or ["?"])[-1], ok, p.stderr
want, _, _ = run({})
gpu = {"SQUID_GPU_INFLATE": "1", "SQUID_INGEST_TIMING": "1"}
bad = 0
for i in range(reps):
    for extra in ({}, {"SQUID_TOK_CAP_MB": "0"}, {"SQUID_TOK_CAP_MB": "0", "SQUID_IL_DEPTH": "3"}, {"SQUID_RESOLVE_GLOBAL": "0", "SQUID_TOK_CAP_MB": "0"}):
        got, ok, err = run(dict(gpu, **extra))
        if got != want or not ok:
            bad += 1
            print("MISMATCH" if got != want else "NOT THE GPU PATH", extra, file=sys.stderr)
            print("\n".join(l for l in err.splitlines() if "ingest" in l or "error" in l.lower() or "squid" in l.lower())[-1500:], file=sys.stderr)
print(f"{reps} rounds, {bad} bad runs")

#!/usr/bin/env python3
"""repeat the GPU reader on a small file under several switch sets and shards, in fresh processes: every run must end normally, take the GPU
path (rc 0) and give the same record arrays as the host reader.  usage: reader_flake.py <prefix> [rounds]"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
pre = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
code = ("import sys, json, hashlib; sys.path.insert(0, %r); import squid_amd\n"
        "ctx = squid_amd.Context(CTX); ctx.load(sys.argv[1], sys.argv[2], shard=SHARD); r = ctx.records()\n"
        "print(json.dumps({k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in r.items()}))") % str(ROOT)
def run(env, shard="None", ctx=""):
    p = subprocess.run([sys.executable, "-c", code.replace("SHARD", shard).replace("CTX", ctx), pre + ".bam", pre + ".chim.bam"], env=dict(os.environ, **env), capture_output=True, text=True)
    ok = p.returncode == 0 and (not env or ("GPU inflate+parse path" in p.stderr and "(rc 0)" in p.stderr))
    return (p.stdout.strip().splitlines() or ["?"])[-1], ok, p.returncode, p.stderr
names, _ = squid_amd.read_header(pre + ".bam")
n = len(names)
shards = [("None", "")] + [(s, f"rank={r}, world_size=3") for r, s in enumerate(("(0, 1)", f"(1, {n - 1})", f"({n - 1}, {n})"))]
want = {sh: run({}, sh, cx)[0] for sh, cx in shards}
gpu = {"SQUID_GPU_INFLATE": "1", "SQUID_INGEST_TIMING": "1"}
bad = runs = 0
for i in range(reps):
    for sh, cx in shards:
        for extra in ({}, {"SQUID_TOK_CAP_MB": "0"}, {"SQUID_NO_BAI": "1"}):
            got, ok, rc, err = run(dict(gpu, **extra), sh, cx)
            runs += 1
            if got != want[sh] or not ok:
                bad += 1
                print(f"BAD rc={rc} shard={sh} {extra}: {'mismatch' if got != want[sh] else 'not the GPU path / died'}", file=sys.stderr)
                print("\n".join(err.splitlines()[-12:]), file=sys.stderr)
print(f"{runs} runs, {bad} bad")

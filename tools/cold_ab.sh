#!/bin/bash
# cold `squid` runs on C3, the GPU left alone for 4 s in front of each; wall clock by python, phases from SQUID_TIMING.  Usage: cold_ab.sh "ENV=..." "ENV=..."
B=build; W=/tmp/sqprobe; mkdir -p $W
[ -f $W/c3.bam ] || $B/gen_synth_bam --config C3 --out $W/c3 --threads 64 > /dev/null
for e in "$@"; do
  for i in 1 2 3; do
    sleep 4
    python3 - "$e" <<'PY'
import os, subprocess, sys, time
env = dict(os.environ, SQUID_TIMING="1", SQUID_INGEST_TIMING="1")
for kv in sys.argv[1].split():
    k, v = kv.split("=", 1); env[k] = v
t0 = time.perf_counter()
p = subprocess.run(["build/squid", "-b", "/tmp/sqprobe/c3.bam", "-c", "/tmp/sqprobe/c3.chim.bam", "-o", "/tmp/sqprobe/out"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
w = time.perf_counter() - t0
keep = [l for l in p.stderr.splitlines() if l.startswith("squid:") or "all " in l and "batches through" in l or "set-up" in l]
print(f"[{sys.argv[1]}] wall {w:.3f} s rc {p.returncode}")
for l in keep[-14:]: print("   ", l)
PY
  done
done

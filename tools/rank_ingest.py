#!/usr/bin/env python3
"""the ingest of one rank's chromosome shard, alone, with the reader's timeline (SQUID_INGEST_TIMING): rank_ingest.py <prefix> <world> <rank> [reps]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SQUID_INGEST_TIMING"] = "1"
import squid_amd
from squid_amd.dist import plan_shards
pre, w, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
_, lens = squid_amd.read_header(f"{pre}.bam")
plan = plan_shards(lens, w)
print("shard", plan[r], file=sys.stderr)
with squid_amd.Context(rank=r, world_size=w) as c:
    for it in range(reps):
        squid_amd.drop_file_cache(); c.clear_records()
        t0 = time.perf_counter(); c.load(f"{pre}.bam", f"{pre}.chim.bam", threads=max(8, 256 // w), shard=plan[r])
        print(f"== rank {r} of {w}, load {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, {c.counts()['n_concordant']} records", file=sys.stderr, flush=True)

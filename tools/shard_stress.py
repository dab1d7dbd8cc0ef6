#!/usr/bin/env python3
"""Randomised check of the chromosome-sharded mode against the unsharded run (GPU box): random seeds, sizes, junction
counts, world sizes and contiguous plans (empty shards included).  usage: tools/shard_stress.py [cases]"""
import random
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import squid_amd  # noqa: E402
import shard_check as sc  # noqa: E402
from squid_amd.dist import VirtualWorld  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(2026)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for case in range(cases):
        cfg = rng.choice(["T2", "T2", "C3"])
        seed = rng.randrange(1, 10**6)
        records = rng.choice([20000, 50000]) if cfg == "T2" else rng.choice([100000, 250000])
        tsv = rng.choice([4, 10, 25]) if cfg == "T2" else rng.choice([20, 80, 200])
        pre = Path(td) / f"s{case}"
        subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", cfg, "--seed", str(seed), "--records", str(records), "--tsv", str(tsv), "--out", str(pre)], stdout=subprocess.DEVNULL)
        try:
            ref = sc.unsharded(pre)
        except squid_amd.SquidError as e:
            print(f"case {case}: {cfg} seed {seed}: unsharded run stops ({str(e)[:60]}), skipped")
            continue
        names, lens = squid_amd.read_header(f"{pre}.bam")
        world = rng.randrange(2, 7)
        cuts = sorted(rng.randrange(0, len(lens) + 1) for _ in range(world - 1))
        plan = list(zip([0] + cuts, cuts + [len(lens)]))
        ctxs = [squid_amd.Context(rank=r, world_size=world) for r in range(world)]
        try:
            for r, c in enumerate(ctxs):
                c.load(f"{pre}.bam", f"{pre}.chim.bam", shard=plan[r])
            vw = VirtualWorld(ctxs)
            vw.build_graph()
            orders = [c.order() for c in ctxs]
            svs = vw.call_sv()
            ok = True
            for r, c in enumerate(ctxs):
                out = {"stages": [c.graph(s) for s in range(6)], "orders": orders[r], "sv": svs[r], "bp": c.breakpoints()}
                for s in range(6):
                    ok &= sc.strip(ref["stages"][s]) == sc.strip(out["stages"][s])
                ok &= ref["orders"] == out["orders"] and ref["sv"] == out["sv"] and ref["bp"] == out["bp"]
            print(f"case {case}: {cfg} seed {seed} records {records} tsv {tsv} world {world} plan {plan} exchanges {vw.exchanges}: {'OK' if ok else 'MISMATCH'}")
            bad += 0 if ok else 1
        except squid_amd.SquidError as e:
            print(f"case {case}: {cfg} seed {seed} world {world} plan {plan}: ERROR {e}")
            bad += 1
        finally:
            for c in ctxs:
                c.close()
print("failures:", bad)
sys.exit(1 if bad else 0)

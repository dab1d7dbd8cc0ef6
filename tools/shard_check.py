#!/usr/bin/env python3
"""Chromosome-sharded run (virtual ranks on one GPU) against the unsharded run of the same files.
usage: tools/shard_check.py <prefix> [world ...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd  # noqa: E402
from squid_amd.dist import VirtualWorld, plan_shards  # noqa: E402


def unsharded(pre):
    with squid_amd.Context() as ctx:
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
        ctx.build_graph()
        out = {"stages": [ctx.graph(s) for s in range(6)], "orders": ctx.order(), "sv": ctx.call_sv(), "bp": ctx.breakpoints(), "counts": ctx.counts()}
    return out


def sharded(pre, world, weights=None):
    names, lens = squid_amd.read_header(f"{pre}.bam")
    plan = plan_shards(weights or lens, world)
    ctxs = [squid_amd.Context(rank=r, world_size=world) for r in range(world)]
    try:
        for r, c in enumerate(ctxs):
            c.load(f"{pre}.bam", f"{pre}.chim.bam", shard=plan[r])
        vw = VirtualWorld(ctxs)
        t0 = time.time()
        vw.build_graph()
        t1 = time.time()
        orders = [c.order() for c in ctxs]
        svs = vw.call_sv()
        outs = [{"stages": [c.graph(s) for s in range(6)], "orders": orders[r], "sv": svs[r], "bp": c.breakpoints(), "counts": c.counts(), "timing": c.timing()} for r, c in enumerate(ctxs)]
        return plan, outs, vw, t1 - t0
    finally:
        for c in ctxs:
            c.close()


def strip(g):  # node support/depth are exact only in exact-depth mode: compare the structure
    return {"nodes": [(n[0], n[1], n[2], n[5]) for n in g["nodes"]], "edges": g["edges"]}


def compare(ref, out, what):
    bad = []
    for s in range(6):
        if strip(ref["stages"][s]) != strip(out["stages"][s]):
            a, b = strip(ref["stages"][s]), strip(out["stages"][s])
            bad.append(f"stage {s}: nodes {len(a['nodes'])} vs {len(b['nodes'])}, edges {len(a['edges'])} vs {len(b['edges'])}")
            if a["nodes"] != b["nodes"]:
                for i, (x, y) in enumerate(zip(a["nodes"], b["nodes"])):
                    if x != y:
                        bad.append(f"   first node diff at {i}: {x} vs {y}")
                        break
            else:
                for i, (x, y) in enumerate(zip(a["edges"], b["edges"])):
                    if x != y:
                        bad.append(f"   first edge diff at {i}: {x} vs {y}")
                        break
    for k in ("orders", "sv", "bp"):
        if ref[k] != out[k]:
            bad.append(f"{k} differ")
    print(f"[{what}] " + ("OK" if not bad else "MISMATCH\n  " + "\n  ".join(bad)))
    return not bad


if __name__ == "__main__":
    pre = sys.argv[1]
    worlds = [int(x) for x in sys.argv[2:]] or [2]
    ref = unsharded(pre)
    print("unsharded: nodes", len(ref["stages"][0]["nodes"]), "edges", len(ref["stages"][0]["edges"]), "sv", len(ref["sv"]), ref["counts"])
    ok = True
    for w in worlds:
        plan, outs, vw, dt = sharded(pre, w)
        print(f"world {w}: plan {plan} exchanges {vw.exchanges} bytes {vw.bytes} records/rank {[o['counts']['n_concordant'] for o in outs]}")
        for r, o in enumerate(outs):
            ok &= compare(ref, o, f"W={w} rank {r}")
    sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Projected strong scaling of the chromosome-sharded mode from virtual ranks on ONE GPU: the ranks of a VirtualWorld
run one after the other, so a step of a real W-GPU run would take about the slowest rank's ingest of its own byte range (each rank measured
alone here: on a real node the ranks share the host's cores and memory, not the GPU or its PCIe link) plus the sum over the phases between
exchanges of the slowest rank's time (RCCL latency of the 5 small all-gathers not included).  Round 6: the ingest is part of the model.
usage: tools/shard_project.py <prefix> [world ...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import squid_amd  # noqa: E402
from squid_amd.dist import VirtualWorld, plan_shards  # noqa: E402

pre = sys.argv[1]
worlds = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
steps = 3


def one_step_unsharded(ctx):
    ctx.reset(); ctx.build_graph(); ctx.order(); return ctx.sv_text()


with squid_amd.Context() as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam")
    ing = []
    for _ in range(3):
        squid_amd.drop_file_cache(); ctx.clear_records()
        t0 = time.perf_counter(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=256); ing.append(time.perf_counter() - t0)
    base_ingest = sorted(ing)[1]
    n = ctx.counts()["n_concordant"]
    one_step_unsharded(ctx)
    t0 = time.perf_counter()
    for _ in range(steps):
        ref = one_step_unsharded(ctx)
    base = (time.perf_counter() - t0) / steps
print(f"unsharded: {n} records, ingest from the page cache {base_ingest * 1e3:.1f} ms, graph pass {base * 1e3:.2f} ms -> step {(base_ingest + base) * 1e3:.1f} ms")
_, lens = squid_amd.read_header(f"{pre}.bam")
for w in worlds:
    plan = plan_shards(lens, w)
    ctxs = [squid_amd.Context(rank=r, world_size=w) for r in range(w)]
    try:
        shard_ingest = []
        for r, c in enumerate(ctxs):
            c.load(f"{pre}.bam", f"{pre}.chim.bam", shard=plan[r])
            best = 1e9
            for _ in range(2):  # (again, warm: what a timed step of bench.py sees)
                squid_amd.drop_file_cache(); c.clear_records()
                t0 = time.perf_counter(); c.load(f"{pre}.bam", f"{pre}.chim.bam", threads=max(8, 256 // w), shard=plan[r]); best = min(best, time.perf_counter() - t0)
            shard_ingest.append(best)
        proj = []
        for it in range(steps + 1):
            for c in ctxs:
                c.reset()
            vw = VirtualWorld(ctxs)
            vw.build_graph()
            t0 = time.perf_counter()
            od = []
            for c in ctxs:
                t1 = time.perf_counter(); c.order(); od.append(time.perf_counter() - t1)
            rows = vw.call_sv()
            if it:
                proj.append(vw.projected_s + max(od))
                last_phases = vw.phases + [("order", max(od))]
        ctxs[0].call_sv = lambda: rows[0]
        same = squid_amd.Context.sv_text(ctxs[0]) == ref
        p = sum(proj) / len(proj)
        shard_n = [c.counts()["n_concordant"] for c in ctxs]
        print("   phases (ms): " + ", ".join(f"{n.replace('_step','')} {t*1e3:.2f}" for n, t in last_phases))
        print("   ingest per rank (ms, each alone on the GPU): " + " ".join(f"{t * 1e3:.0f}" for t in shard_ingest))
        step = max(shard_ingest) + p
        print(f"W={w}: projected graph pass {p * 1e3:.2f} ms ({base / p:.2f}x), slowest ingest {max(shard_ingest) * 1e3:.1f} ms ({base_ingest / max(shard_ingest):.2f}x) -> step {step * 1e3:.1f} ms = {(base_ingest + base) / step:.2f}x  "
              f"(largest shard {max(shard_n) / n:.1%} of the records, {vw.exchanges} exchanges, {vw.bytes} B, sv identical: {same})")
    finally:
        for c in ctxs:
            c.close()

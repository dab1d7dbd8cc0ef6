#!/usr/bin/env python3
"""summary of a rocprofv3 kernel_trace.csv: for the LAST ingest in the trace (dispatches after the last long gap), per kernel
name the dispatch count, summed duration and the union of its busy intervals, plus how much of the wall time had 1, 2, 3+
different kernel names running at once."""
import csv, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "").replace("sq::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# the last ingest: back from the end until a gap of more than 50 ms without any kernel
cut = 0
last_end = rows[0][1]
for i, (s, e, n) in enumerate(rows):
    if s - last_end > 50_000_000:
        cut = i
    last_end = max(last_end, e)
rows = rows[cut:]
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
print(f"window {(t1 - t0) / 1e6:.1f} ms, {len(rows)} dispatches")
per = defaultdict(list)
for s, e, n in rows:
    per[n].append((s, e))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for n, iv in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:14]:
    print(f"{n[:44]:44s} n={len(iv):4d} sum={sum(e - s for s, e in iv) / 1e6:8.1f} ms  busy={union(iv) / 1e6:8.1f} ms  first={(min(s for s, _ in iv) - t0) / 1e6:7.1f} last_end={(max(e for _, e in iv) - t0) / 1e6:7.1f}")
ev = []
for s, e, n in rows:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active = defaultdict(int); hist = defaultdict(int); prev = t0
for t, d, n in ev:
    k = sum(1 for v in active.values() if v > 0)
    hist[min(k, 4)] += t - prev; prev = t
    active[n] += d
print("distinct kernels running at once:", {k: f"{v / 1e6:.1f} ms" for k, v in sorted(hist.items())})
if len(sys.argv) > 2 and sys.argv[2] == "--list":  # every dispatch of more than 0.3 ms, in start order
    for s, e, n in rows:
        if e - s > 300_000:
            print(f"  {(s - t0) / 1e6:8.2f} .. {(e - t0) / 1e6:8.2f}  {(e - s) / 1e6:7.2f} ms  {n[:40]}")

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of staged C3 steps: per kernel count / mean / sum / union over the LAST step, and which
kernels were running in every 10 ms slice of it.  Usage: ingest_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sq::", "").replace("void ", "")))
rows.sort()
starts = [s for s, e, k in rows if k.startswith("k_inflate_tok2") or k.startswith("k_inflate_spec")]
# steps are separated by gaps > 20 ms between token launches... take the last run of launches
cut = starts[0]
for a, b in zip(starts, starts[1:]):
    if b - a > 60e6: cut = b
last = [r for r in rows if r[0] >= cut - 2e6]
t0 = last[0][0]; t1 = max(e for s, e, k in last)
print(f"last step: {len(last)} launches over {(t1 - t0) / 1e6:.1f} ms")
by = collections.defaultdict(list)
for s, e, k in last: by[k].append((s, e))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"{'kernel':32s} {'n':>5s} {'mean ms':>9s} {'sum ms':>9s} {'union ms':>9s} {'first':>7s} {'last end':>8s}")
for k, iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:14]:
    print(f"{k[:32]:32s} {len(iv):5d} {sum(e - s for s, e in iv) / len(iv) / 1e6:9.3f} {sum(e - s for s, e in iv) / 1e6:9.2f} {union(iv) / 1e6:9.2f} {(min(s for s, e in iv) - t0) / 1e6:7.1f} {(max(e for s, e in iv) - t0) / 1e6:8.1f}")
tokname = next((k for k in by if k.startswith("k_inflate_spec") or k.startswith("k_inflate_tok2")), "k_inflate_tok2<false>")
print("token launches (start, end ms):", " ".join(f"{(s - t0) / 1e6:.1f}-{(e - t0) / 1e6:.1f}" for s, e in sorted(by.get(tokname, []))))
tail0 = max(e for s, e in by.get(tokname, [(t0, t0)]))
print(f"after the last token pass ended (+{(tail0 - t0) / 1e6:.1f} ms): " + " ".join(f"{k[:18]}@{(s - tail0) / 1e6:.2f}+{(e - s) / 1e6:.2f}" for s, e, k in last if s >= tail0 and e - s > 30e3))
for name in by:
    if name.startswith("k_lz_resolve"):
        print(name, "launches:", " ".join(f"{(s - t0) / 1e6:.1f}-{(e - t0) / 1e6:.1f}" for s, e in sorted(by[name])))
for name in ("k_parse_records", "k_rec_sync"):
    for k in by:
        if k.startswith(name): print(k, "launches:", " ".join(f"{(s - t0) / 1e6:.1f}-{(e - t0) / 1e6:.1f}" for s, e in sorted(by[k])))

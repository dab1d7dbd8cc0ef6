#!/usr/bin/env python3
"""Last from-file step of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/file_steps.py: when the file's bytes arrived, when the
token passes ran, what the GPU did behind the last copy.  Usage: file_trace.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv, sys
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sq::", "").replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
M = []
for r in csv.DictReader(open(sys.argv[2])):
    d = r.get("Direction", "")
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    M.append((s_, e_, d, (8 << 20) if ("HOST_TO_DEVICE" in d and e_ - s_ > 80e3) else 0))  # (the trace has no sizes: a host-to-device copy of more than 80 us is taken for one of the feeder's 8 MiB pieces)
K.sort(); M.sort()
tok = [(s, e) for s, e, k in K if k.startswith("k_inflate_spec")]
cut = tok[0][0]
for (a, _), (b, _) in zip(tok, tok[1:]):
    if b - a > 100e6: cut = b
big = [m for m in M if m[3] >= (1 << 20) and m[0] >= cut - 30e6]  # the file pieces of the last step
t0 = min(big[0][0], cut)
last = [k for k in K if k[0] >= t0]
t1 = max(e for s, e, k in last)
print(f"last step: {(t1 - t0) / 1e6:.1f} ms from the first piece's copy to the last kernel; {len(big)} copies of {sum(m[3] for m in big) / 1e9:.2f} GB, the last ends at {(big[-1][1] - t0) / 1e6:.1f} ms")
tot = 0
marks = []
for s, e, d, n in big:
    tot += n
    marks.append((e, tot))
def arrived(t): 
    a = 0
    for e, c in marks:
        if e <= t: a = c
        else: break
    return a
print("token passes (start-end ms | GB of file arrived at start):")
print("  " + "  ".join(f"{(s - t0) / 1e6:.1f}-{(e - t0) / 1e6:.1f}|{arrived(s) / 1e9:.2f}" for s, e in tok if s >= t0))
iv = sorted((s, e) for s, e, k in last)
idle = 0; ce = iv[0][1]; gaps = []
for s, e in iv[1:]:
    if s > ce:
        idle += s - ce
        if s - ce > 300e3: gaps.append(((ce - t0) / 1e6, (s - ce) / 1e6))
    ce = max(ce, e)
print(f"no kernel running: {idle / 1e6:.1f} ms in all; gaps > 0.3 ms (at, length): " + " ".join(f"{a:.1f}+{g:.1f}" for a, g in gaps))
lastcopy = big[-1][1]
print("behind the last copy: " + " ".join(f"{k[:16]}@{(s - lastcopy) / 1e6:.2f}+{(e - s) / 1e6:.2f}" for s, e, k in last if e > lastcopy and e - s > 100e3)[:3000])

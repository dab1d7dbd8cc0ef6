#!/usr/bin/env python3
"""does the container's CPU quota throttle the process during a step?  (cgroup v2 cpu.stat before / after every step)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd

def stat():
    d = {}
    try:
        for l in open("/sys/fs/cgroup/cpu.stat"):
            k, v = l.split(); d[k] = int(v)
    except OSError:
        pass
    return d

args = [a for a in sys.argv[1:] if not a.startswith("--")]
staged = "--staged" in sys.argv
if "--keep" in sys.argv:
    squid_amd.keep_host_memory()
wl = args[0] if args else "C3"
pre = {"C3": "/tmp/squid_bench/C3_s20180003", "C5": "/tmp/squid_bench/C5_s20180005"}[wl]
if not os.path.exists(pre + ".bam"):  # (a fresh box: make the sample the way bench.py does)
    import subprocess
    os.makedirs(os.path.dirname(pre), exist_ok=True)
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", wl, "--seed", pre.rsplit("_s", 1)[1], "--out", pre, "--threads", "128"], stdout=subprocess.DEVNULL)
kw = dict(min_edge_weight=1, max_allowed_degree=50) if wl == "C5" else {}
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
with squid_amd.Context(**kw) as ctx:
    ctx.keep_stage_graphs(False)
    if staged:
        ctx.stage_bam(f"{pre}.bam")
    for it in range(5):
        squid_amd.drop_file_cache(); ctx.clear_records()
        a = stat(); t0 = time.perf_counter()
        ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=256)
        t1 = time.perf_counter()
        ctx.build_graph(); ctx.order_sizes(); ctx.sv_text_fast()
        t2 = time.perf_counter(); b = stat()
        print(f"step {it}: load {1e3*(t1-t0):.1f} ms, pass {1e3*(t2-t1):.1f} ms; cpu used {(b.get('usage_usec',0)-a.get('usage_usec',0))/1e3:.0f} ms, "
              f"periods {b.get('nr_periods',0)-a.get('nr_periods',0)}, throttled {b.get('nr_throttled',0)-a.get('nr_throttled',0)} times for {(b.get('throttled_usec',0)-a.get('throttled_usec',0))/1e3:.0f} ms")

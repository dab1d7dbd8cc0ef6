#!/usr/bin/env python3
"""time k_depth2 cut short (SQUID_D2_ABLATE: 1 = the loads of phase 1 alone, 2 = without the sums of the one-node tiles; results unused)"""
import os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
rec = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
pre = Path(tempfile.gettempdir()) / f"pt_C3_{rec}_0"
if not Path(f"{pre}.bam").exists():
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", "C3", "--records", str(rec), "--out", str(pre), "--threads", str(os.cpu_count() or 8)], stdout=subprocess.DEVNULL)
with squid_amd.Context() as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
    for ab in (0, 1, 2, 0):
        os.environ["SQUID_D2_ABLATE"] = str(ab)
        ms = []
        for it in range(4):
            ctx.reset()
            try:
                ctx.build_graph()
            except Exception:
                pass
            t = ctx.timing().get("k_depth2")
            if t and it: ms.append(t["ms"] / max(1, t["launches"]))
        print(f"ablate {ab}: k_depth2 {sum(ms) / max(1, len(ms)):.4f} ms")

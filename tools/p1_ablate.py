#!/usr/bin/env python3
"""time k_pass1 cut off after each of its sections (SQUID_P1_ABLATE; results are wrong, timing only)"""
import os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import squid_amd
rec = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
pre = Path(tempfile.gettempdir()) / f"pt_C3_{rec}"
if not Path(f"{pre}.bam").exists():
    subprocess.check_call([str(ROOT / "build" / "gen_synth_bam"), "--config", "C3", "--records", str(rec), "--out", str(pre), "--threads", str(os.cpu_count() or 8)], stdout=subprocess.DEVNULL)
with squid_amd.Context() as ctx:
    ctx.load(f"{pre}.bam", f"{pre}.chim.bam", threads=16)
    for ab in [int(x) for x in os.environ.get('P1_ABLATE_LIST', '0,1,2,4,8,16,32,64,128,126,254').split(',')]:
        os.environ["SQUID_P1_ABLATE"] = str(ab)
        ms = []
        for it in range(4):
            ctx.reset()
            try:
                ctx.build_graph()
            except Exception as e:
                pass
            t = ctx.timing().get("k_pass1w") or ctx.timing().get("k_pass1")
            if t and it: ms.append(t["ms"] / max(1, t["launches"]))
        print(f"ablate {ab}: k_pass1 {sum(ms) / max(1, len(ms)):.4f} ms")

#!/usr/bin/env python3
"""the record parse at several read lengths (records of 100 / 150 / 250 bases: 230 / 310 / 460 bytes each): time of k_parse_records per million records.
The records of a workgroup are staged in LDS when they fit: this shows what a workgroup that does not fit costs."""
import os, random, struct, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ["SQUID_GPU_INFLATE"] = "1"
import bamwriter as bw
import squid_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
chim = [bw.record("q1", 0, 5000, 255, 0x1 | 0x40, "60M40S"), bw.record("q1", 1, 7000, 255, 0x1 | 0x40 | 0x100, "60H40M"), bw.record("q1", 1, 7300, 255, 0x1 | 0x80 | 0x10, "100M")]
for L in (100, 150, 250):
    rng = random.Random(L)
    one = lambda i: bw.record(f"read{i}", 0, 1000 + i // 4, 255, 0x1 | 0x2 | 0x20 | 0x40, f"{L}M", 0, 1000 + i // 4 + 200, seq="".join(rng.choice("ACGT") for _ in range(L)), qual=[rng.randrange(20, 40) for _ in range(L)], tags=b"NHC\x01HIC\x01ASC\x62nMC\x00")
    recs = [one(i) for i in range(2000)]
    recs = (recs * (n // 2000 + 1))[:n]  # (the writer is Python: a block of records repeated)
    pre = f"/tmp/parse_probe_{L}"
    bw.write_bam(f"{pre}.bam", (("chrA", 10000000), ("chrB", 50000)), recs)
    bw.write_bam(f"{pre}.chim.bam", (("chrA", 10000000), ("chrB", 50000)), chim, sort_order="unsorted")
    for kb in (os.environ.get("PROBE_KB", "0").split(",")):
        os.environ.pop("SQUID_PARSE_LDS_KB", None)
        if kb != "0": os.environ["SQUID_PARSE_LDS_KB"] = kb
        import subprocess
        code = ("import sys; sys.path.insert(0, %r); import squid_amd\n"
                "with squid_amd.Context() as ctx:\n"
                "    seen = []\n"
                "    for it in range(6):\n"
                "        ctx.clear_records(); ctx.load(sys.argv[1] + '.bam', sys.argv[1] + '.chim.bam')\n"
                "        k = ctx.timing().get('k_parse_records', {}); seen.append(k.get('ms', 0) / max(1, k.get('launches', 1)))\n"
                "    print(' '.join('%%.3f' %% x for x in seen), ctx.counts()['n_concordant'])\n") % str(ROOT)
        out = subprocess.run([sys.executable, "-c", code, pre], capture_output=True, text=True)
        print(f"reads of {L} bases ({len(recs[0])} bytes per record), staging {kb if kb != '0' else 'auto'} KB: k_parse_records ms per launch over six loads: {out.stdout.strip()} {out.stderr[-300:] if out.returncode else ''}", flush=True)

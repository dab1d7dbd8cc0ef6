#!/usr/bin/env python3
"""HIP path against an oracle dump made elsewhere (the oracle is slow on dense-graph inputs; no GPU needed for it).
usage: tools/compare_dump.py <bam prefix> <dump dir> <oracle _sv.txt> [param=value ...]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import squid_amd  # noqa: E402
import test_gpu_parity as T  # noqa: E402

pre, dump, sv = sys.argv[1], Path(sys.argv[2]), Path(sys.argv[3])
params = {k: (float(v) if "." in v else int(v)) for k, v in (a.split("=") for a in sys.argv[4:])}
with squid_amd.Context(**params) as ctx:
    t0 = time.time(); ctx.load(f"{pre}.bam", f"{pre}.chim.bam"); t1 = time.time()
    ctx.build_graph(); t2 = time.time()
    ctx.order(); t3 = time.time()
    T._compare(ctx, dump, sv, depth_exact=False)
    print(f"PARITY OK  records={ctx.counts()['n_concordant']} load={t1 - t0:.2f}s graph={(t2 - t1) * 1e3:.1f}ms order={(t3 - t2) * 1e3:.1f}ms")
    for step in range(2):
        ctx.reset(); a = time.time(); ctx.build_graph(); b = time.time(); ctx.order(); c2 = time.time(); ctx.sv_text(); d = time.time()
    print(f"steady: graph={(b - a) * 1e3:.1f}ms order={(c2 - b) * 1e3:.1f}ms sv={(d - c2) * 1e3:.1f}ms")
    for k, v in sorted(ctx.timing().items(), key=lambda kv: -kv[1]["ms"])[:12]:
        print(f"   {k:28s} {v['ms']:.3f} ms")

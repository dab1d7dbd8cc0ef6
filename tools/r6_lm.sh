#!/bin/bash
# how the two resolves take a file of long matches (tests' long-match data, scaled up): tok_bench on it
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
python3 - <<'PY'
import random, sys
sys.path.insert(0, 'tests')
import bamwriter as bw
rng = random.Random(11)
recs = []
for i in range(400000):
    p = 1000 + i // 400
    kind = (i // 500) % 3
    if kind == 0: tags = b"NHC\x01" + b"ZZZ" + b"ACGTTGCA" * 30 + b"\0"
    elif kind == 1: tags = b"NHC\x01" + b"ZZZ" + bytes([65 + i % 3]) * rng.randrange(40, 700) + b"\0"
    else: tags = b"NHC\x01" + b"ZZZ" + bytes(rng.choice(b"ACGTNacgtn0123456789") for _ in range(rng.randrange(10, 300))) + b"\0"
    recs.append(bw.record("same" if kind == 0 else f"r{i}", 0, p, 255, 0x1 | 0x2 | 0x20 | 0x40, "100M", 0, p + 200, tags=tags))
bw.write_bam("/tmp/longm.bam", (("chrA", 100000), ("chrB", 50000)), recs)
PY
ls -la /tmp/longm.bam
for S in 0 1; do echo "SQUID_RESOLVE_STAGED=$S: $(SQUID_RESOLVE_STAGED=$S timeout 120 python tools/tok_bench.py /tmp/longm.bam 2000 3 25610 2>&1 | grep variant | sed 's/.*| resolve/resolve/')"; done

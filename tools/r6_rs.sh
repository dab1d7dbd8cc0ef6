#!/bin/bash
# the staged resolve (k_lz_resolve5) alone against k_lz_resolve3 (tools/tok_bench.py compares every block with zlib), then in the reader
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p /tmp/squid_bench gpurun_out/r6rs
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for S in 0 1; do echo "SQUID_RESOLVE_STAGED=$S: $(SQUID_RESOLVE_STAGED=$S timeout 120 python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 25610 2>&1 | grep variant | sed 's/.*| resolve/resolve/')"; done | tee gpurun_out/r6rs/alone.txt
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli --steps 6 --staged-steps 6 > gpurun_out/r6rs/$tag.json 2> gpurun_out/r6rs/$tag.err; python3 - gpurun_out/r6rs/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s} file {d['ms_per_step']:.1f} ms  staged {d.get('staged_ms_per_step', 0):.1f} ms  " + " ".join(f"{k[2:12]} {v['us_per_launch']:.0f}us/{v['busy_ms_per_step']:.0f}" for k, v in d['ingest_kernels'].items()) + f"  parity_failures {d.get('parity_failures')}")
PY
}
run r3 SQUID_RESOLVE_STAGED=0
run r5 SQUID_RESOLVE_STAGED=1
run r3b SQUID_RESOLVE_STAGED=0
run r5b SQUID_RESOLVE_STAGED=1

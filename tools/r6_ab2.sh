#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6ab
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli --steps 8 --staged-steps 6 > gpurun_out/r6ab/$tag.json 2> gpurun_out/r6ab/$tag.err; python3 - gpurun_out/r6ab/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:14s} file {d['ms_per_step']:.1f} ms {d.get('ms_each')} staged {d.get('staged_ms_per_step', 0):.1f} ms  launches {d['ingest_kernels']['k_inflate_spec']['launches_per_step']}")
PY
}
run taper1 A=1
run taper0 SQUID_TOK_TAPER=0
run taper1b A=1
run taper0b SQUID_TOK_TAPER=0

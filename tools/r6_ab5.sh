#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6ab
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli --steps 6 --staged-steps 6 > gpurun_out/r6ab/$tag.json 2> gpurun_out/r6ab/$tag.err; python3 - gpurun_out/r6ab/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s} file {d['ms_per_step']:.1f} ms  staged {d.get('staged_ms_per_step', 0):.1f} ms  " + " ".join(f"{k[2:12]} {v['us_per_launch']:.0f}us/{v['busy_ms_per_step']:.0f}" for k, v in d['ingest_kernels'].items()))
PY
}
run base A=1
run cap384 SQUID_TOK_CAP_MB=384
run cap640 SQUID_TOK_CAP_MB=640
run d4 SQUID_IL_DEPTH=4
run d4cap384 SQUID_IL_DEPTH=4 SQUID_TOK_CAP_MB=384
run base2 A=1

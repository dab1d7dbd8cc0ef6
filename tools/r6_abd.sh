#!/bin/bash
# A/B on the dense config (and the --bwa line): this tree's library against build/ab/lib_old.so (the tree before the reader changes of the round's second half)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6abd
run() { tag=$1; shift; env "$@" python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-bwa --no-cold-cli --resident-steps 1 --staged-steps 0 > gpurun_out/r6abd/$tag.json 2> gpurun_out/r6abd/$tag.err; python3 - gpurun_out/r6abd/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:8s} C5 from file {d['ms_per_step']:.0f} ms {d.get('ms_each')}")
PY
}
run new A=1
run old SQUID_LIB=$PWD/build/ab/lib_old.so
run new2 A=1
run old2 SQUID_LIB=$PWD/build/ab/lib_old.so

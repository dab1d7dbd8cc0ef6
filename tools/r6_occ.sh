#!/bin/bash
# the resolve alone at several occupancies (SQUID_RESOLVE_LDS pads its LDS)
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p /tmp/squid_bench gpurun_out/r6o
[ -f /tmp/squid_bench/C3.bam ] || build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for L in 0 5120 6800 8192 10240 13600 20480; do
  echo "SQUID_RESOLVE_LDS=$L: $(SQUID_RESOLVE_LDS=$L timeout 120 python tools/tok_bench.py /tmp/squid_bench/C3.bam 16384 3 25610 2>&1 | grep variant | sed 's/.*| resolve/resolve/')"
done | tee gpurun_out/r6o/occ.txt

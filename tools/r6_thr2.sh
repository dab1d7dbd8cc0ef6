#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6thr
for t in new old new old; do
  if [ $t = old ]; then export SQUID_LIB=$PWD/build/ab/lib_old.so; else unset SQUID_LIB; fi
  echo "== $t"; python3 tools/throttle_probe.py C5 2>&1 | grep "^step" | tail -4
done | tee gpurun_out/r6thr/c5_ab.txt

#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh <old.so> <new.so> [pass_timing arguments]
OLD=$1; NEW=$2; shift 2
cp build/libsquid_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp $OLD build/libsquid_hip.so; else cp $NEW build/libsquid_hip.so; fi
    timeout 300 python tools/pass_timing.py "$@" 2>&1 | grep "k_pass1\|k_depth2" | tr "\n" " "; echo " $v"
  done
done
cp /tmp/lib_keep.so build/libsquid_hip.so

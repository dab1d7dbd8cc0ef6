#!/bin/bash
# same-box A/B of the token pass of two builds of the library: tools/tok_ab.sh <old.so> <prefix of a C3 sample>
OLD=$1; PRE=$2
cp build/libsquid_hip.so /tmp/lib_keep.so
for v in old new old new; do
  if [ $v = old ]; then cp $OLD build/libsquid_hip.so; else cp /tmp/lib_keep.so build/libsquid_hip.so; fi
  echo "== $v"
  SQUID_GPU_INFLATE=1 SQUID_TOK_PROF=1 python tools/ingest_once.py $PRE 2>&1 | grep -E "token pass profile|records in|k_inflate_tok2|k_lz_resolve2"
done
cp /tmp/lib_keep.so build/libsquid_hip.so

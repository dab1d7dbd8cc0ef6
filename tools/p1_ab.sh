#!/bin/bash
# per-kernel timings of the graph pass over the resident C3 records for several builds of the library (SQUID_LIB), same box, same records
# usage: tools/p1_ab.sh lib[:ENV=val[,ENV=val]] ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
  echo "==== $lib $envs"
  env SQUID_LIB=$PWD/$lib ${envs//,/ } python3 tools/pass_timing.py --records ${P1_AB_RECORDS:-50000000} --passes 6 2>&1 | grep -E "SV rows|k_pass1|k_tile|k_zfinal|k_depth|k_edges|k_bp2|k_bp_|k_summ|wall_build" | cut -c1-100
done

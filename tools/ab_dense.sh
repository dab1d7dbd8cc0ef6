#!/bin/bash
# A/B of host-side settings on the dense config: load / pass medians of tools/throttle_probe.py, the variants interleaved and repeated
# usage: tools/ab_dense.sh ROUNDS "ENV=.. ENV=.. [--flag]" "..." ...
cd "$(dirname "$0")/.."
R=$1; shift
for r in $(seq $R); do
  for v in "$@"; do
    envs=$(echo "$v" | tr ' ' '\n' | grep = | tr '\n' ' '); flags=$(echo "$v" | tr ' ' '\n' | grep -v = | tr '\n' ' ')
    out=$(env $envs python tools/throttle_probe.py C5 $flags 2>&1 | grep "^step" | tail -4)
    echo "$out" | python -c "
import sys,re,statistics
l=[];p=[];c=[]
for line in sys.stdin:
    m=re.search(r'load ([\d.]+) ms, pass ([\d.]+) ms; cpu used (\d+)',line)
    if m: l.append(float(m.group(1)));p.append(float(m.group(2)));c.append(int(m.group(3)))
print('round $r  %-70s load %7.1f  pass %7.1f  cpu %6d' % ('$v', statistics.median(l), statistics.median(p), statistics.median(c)))"
  done
done

#!/bin/bash
# A/B of builds (SQUID_LIB=...) and switches of the library on staged C3 steps (GPU box); edit the run lines
build/gen_synth_bam --config C3 --out /tmp/c3 --threads 64 > /dev/null
run() { echo "-- $*"; env "$@" python3 tools/staged_steps.py /tmp/c3 7 2>&1 | grep "^== steps"; }
run X=base
for v in scan4 scan8 p64_16 scan4p; do run SQUID_LIB=$PWD/build/ab/lib_$v.so; done
run X=base
for v in scan4 scan8; do echo "-- resident pass $v"; SQUID_LIB=$PWD/build/ab/lib_$v.so python3 tools/pass_timing.py --records 50000000 2>&1 | grep "ms per pass" | cut -c1-60; done
echo "-- resident pass base"; python3 tools/pass_timing.py --records 50000000 2>&1 | grep "ms per pass" | cut -c1-60

#!/bin/bash
# A/B of builds (SQUID_LIB=...) and switches of the library on staged C3 steps (GPU box); edit the run lines
build/gen_synth_bam --config C3 --out /tmp/c3 --threads 64 > /dev/null
run() { echo "-- $*"; env "$@" python3 tools/staged_steps.py /tmp/c3 7 2>&1 | grep "^== steps"; }
run X=base
for v in p16 p32 p16s; do run SQUID_LIB=$PWD/build/ab/lib_$v.so; done
run X=base

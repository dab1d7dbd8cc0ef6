#!/bin/bash
# A/B of builds (SQUID_LIB=...) and switches of the library on staged C3 steps (GPU box); edit the run lines
build/gen_synth_bam --config C3 --out /tmp/c3 --threads 64 > /dev/null
run() { echo "-- $*"; env "$@" timeout 120 python3 tools/staged_steps.py /tmp/c3 9 2>&1 | grep "^== steps"; }
for rep in 1 2 3; do
run X=base
run SQUID_LIB=$PWD/build/ab/lib_lits3.so
done

B=build; W=/tmp/sqprobe; mkdir -p $W
[ -f $W/c3.bam ] || $B/gen_synth_bam --config C3 --out $W/c3 --threads 64 > /dev/null
for d in 8 8 5; do sleep 4; echo "=== depth $d"; SQUID_IL_DEPTH=$d SQUID_TIMING=1 SQUID_INGEST_TIMING=1 $B/squid -b $W/c3.bam -c $W/c3.chim.bam -o $W/out 2>&1 >/dev/null | grep "squid +\|GPU ingest\|ingest /"; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6k /tmp/squid_bench
build/gen_synth_bam --config C3 --seed 20180003 --out /tmp/squid_bench/C3 --threads 32 > /dev/null 2>&1
for v in plain torch torch_omp1; do
  case $v in plain) E="";; torch) E="WITH_TORCH=1";; torch_omp1) E="WITH_TORCH=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1";; esac
  env $E timeout 600 python tools/file_step_timeline.py /tmp/squid_bench/C3 > gpurun_out/r6k/file_$v.log 2>&1
  echo "$v: $(grep '== step' gpurun_out/r6k/file_$v.log | sed 's/== step [0-9]: //' | tr '\n' '|')"
  grep "file pieces queued" gpurun_out/r6k/file_$v.log | tail -2
done

#!/bin/bash
# reader tests, then short bench lines at buffer-set depths 3 and 4
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r6r
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reader or inflate or token or route or damaged or four_million" > gpurun_out/r6r/pytest.log 2>&1; tail -5 gpurun_out/r6r/pytest.log
for d in 3 4; do
  SQUID_IL_DEPTH=$d python bench.py --no-cpu-baseline --no-dense --no-bwa --no-cold-cli > gpurun_out/r6r/bench_d$d.json 2> gpurun_out/r6r/bench_d$d.err
  echo "depth $d bench rc $?"; tail -2 gpurun_out/r6r/bench_d$d.err | cut -c1-300
done

#!/bin/bash
# plumbing check of the N > 1 bench path with the final reader: two and eight processes on ONE GPU (gloo), C3, chromosome shards
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=4 SQUID_DIST_BACKEND=gloo
mkdir -p gpurun_out/r6dry
for N in 2 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-bwa --no-cold-cli > gpurun_out/r6dry/n$N.json 2> gpurun_out/r6dry/n$N.err
  echo "N=$N rc $?"; tail -c 600 gpurun_out/r6dry/n$N.json | head -c 600; echo
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-bwa --no-cold-cli 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1 sha', d.get('sv_sha256'), d['value'])"

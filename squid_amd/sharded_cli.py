"""`squid -b/-c/-o` over several GPUs of one node, one process per GPU, records sharded by chromosome:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m squid_amd.sharded_cli -b sorted.bam -c chimeric.bam -o out/prefix [any other squid flag]

The flags are parsed by the drop-in CLI itself (`build/squid --print-config`, which follows src/Config.cpp:80-230),
every rank ingests the chimeric BAM and the concordant records of its chromosome range, the library carries out its exchanges
itself (include/squid_hip.h, sq_exchange: one RCCL all-gather each, over a communicator made inside the library), and rank 0 writes `<prefix>_sv.txt` exactly as the
single-GPU `squid` does (src/WriteIO.cpp:45-124).  SQUID_DIST_BACKEND=gloo lets the ranks share one GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import squid_amd
from squid_amd.dist import install_native_exchange, plan_shards, shard_weights


def parse_flags(argv: list) -> dict:
    out = subprocess.run([str(squid_amd.BUILD / "squid"), "--print-config", *argv], capture_output=True, text=True, check=True).stdout.split()
    return dict(kv.split("=", 1) for kv in out if "=" in kv)


def main(argv: list) -> int:
    import torch
    import torch.distributed as dist

    cfg = parse_flags(argv)
    if cfg.get("ok") != "1" or not cfg.get("b") or not cfg.get("c"):
        print("Check your argument.")  # src/Config.cpp:227-229 (the sharded runner needs -b and -c)
        return 0
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    backend = os.environ.get("SQUID_DIST_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    params = dict(phred_type=int(cfg["pt"]), max_lowphred_len=int(cfg["pl"]), min_phred=int(cfg["pm"]), min_mapqual=int(cfg["mq"]), concord_dist_pos=int(cfg["dp"]),
                  concord_dist_idx=int(cfg["di"]), min_edge_weight=int(cfg["w"]), discordant_ratio=float(cfg["r"]), max_allowed_degree=int(cfg["a"]))
    shard = None
    if world > 1:
        _, ref_len = squid_amd.read_header(cfg["b"])
        shard = plan_shards(shard_weights(cfg["b"], ref_len), world)[rank]
        params.update(rank=rank, world_size=world)
    with squid_amd.Context(device=local, star_mapq=False, **params) as ctx:
        if world > 1:
            install_native_exchange(ctx, dist, backend)  # sq_exchange: RCCL inside the library, or a gloo all-gather as its transport
        ctx.load(cfg["b"], cfg["c"], threads=max(1, (os.cpu_count() or 8) // world), shard=shard)
        ctx.build_graph()
        ctx.order()
        text = ctx.sv_text()
    if rank == 0:
        Path(cfg["o"] + "_sv.txt").write_text(text)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

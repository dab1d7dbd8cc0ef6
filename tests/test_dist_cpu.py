"""CPU (gloo, world_size 2): the multi-rank plumbing used by `bench.py --gpus N`."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from squid_amd.dist import assign_samples, reduce_timing


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed, units = reduce_timing(1.0 + rank, 1000.0 * (rank + 1), dist, device="cpu")
    q.put((rank, elapsed, units, assign_samples(5, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_timing_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, units, mine in got:
        assert elapsed == 2.0          # max over ranks
        assert units == 3000.0         # sum over ranks
    assert got[0][3] == [0, 2, 4] and got[1][3] == [1, 3]


def test_single_process_passthrough():
    assert reduce_timing(0.5, 10.0, None) == (0.5, 10.0)

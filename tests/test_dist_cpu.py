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


# ---- chromosome-sharded mode: partition plan and the variable-length all-gather that carries the library's exchanges
def test_plan_shards_is_contiguous_balanced_and_complete():
    from squid_amd.dist import plan_shards

    hg38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328,
            107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]
    for world in (1, 2, 3, 4, 8, 30):
        plan = plan_shards(hg38, world)
        assert len(plan) == world
        assert plan[0][0] == 0 and plan[-1][1] == len(hg38)
        for (a, b), (c, d) in zip(plan, plan[1:]):
            assert b == c and a <= b  # contiguous, in order
        loads = [sum(hg38[a:b]) for a, b in plan]
        assert sum(loads) == sum(hg38)
        assert max(loads) >= max(hg38)
        if world == 8:
            assert max(loads) < 1.25 * sum(hg38) / 8
    # optimal among all contiguous splits (brute force on small inputs)
    import itertools
    import random

    rng = random.Random(3)
    for _ in range(50):
        w = [rng.randrange(0, 20) for _ in range(rng.randrange(1, 9))]
        world = rng.randrange(1, 5)
        best = min(max(sum(w[a:b]) for a, b in zip((0,) + cuts, cuts + (len(w),))) for cuts in itertools.combinations_with_replacement(range(len(w) + 1), world - 1))
        plan = plan_shards(w, world)
        assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == len(w)
        assert max(sum(w[a:b]) for a, b in plan) == best, (w, world, plan)
    assert plan_shards([5, 5], 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert plan_shards([], 2) == [(0, 0), (0, 0)]


class _FakeShardCtx:
    """mimics the stage protocol of a sharded sq_ctx (SQ_NEED_EXCHANGE / pack / unpack): every rank contributes a
    payload of a different length per round and checks what comes back"""

    NEED_EXCHANGE = 1

    def __init__(self, rank, world, rounds):
        self.rank, self.world, self.rounds, self.round, self.pending, self.seen = rank, world, rounds, 0, False, []

    def payload(self, rank, rnd):
        return bytes([rank, rnd]) * (1 + 37 * rank + 5 * rnd)

    def step(self):
        if self.pending:
            raise RuntimeError("exchange skipped")
        if self.round == self.rounds:
            return 0
        self.pending = True
        return self.NEED_EXCHANGE

    def exchange_pack(self):
        return self.payload(self.rank, self.round)

    def exchange_unpack(self, parts):
        assert parts == [self.payload(r, self.round) for r in range(self.world)]
        self.seen.append(sum(len(x) for x in parts))
        self.pending = False
        self.round += 1


def _shard_worker(rank, world, port, q):
    from squid_amd.dist import TorchExchange

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = TorchExchange(dist, device="cpu")
    ctx = _FakeShardCtx(rank, world, rounds=4)
    while ctx.step() == ctx.NEED_EXCHANGE:
        ctx.exchange_unpack(ex(ctx.exchange_pack()))
    empty = ex(b"" if rank == 0 else b"x")  # an empty contribution is legal
    q.put((rank, ctx.seen, ex.calls, [len(x) for x in empty]))
    dist.barrier()
    dist.destroy_process_group()


def test_variable_length_allgather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 500
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and len(got[0][1]) == 4
    assert got[0][2] == got[1][2] == 5
    assert got[0][3] == got[1][3] == [0, 1]


def test_virtual_world_lockstep():
    from squid_amd.dist import VirtualWorld

    class C(_FakeShardCtx):
        def build_graph_step(self):
            return self.step()

    ctxs = [C(r, 3, rounds=2 + 0) for r in range(3)]
    vw = VirtualWorld(ctxs)
    vw.build_graph()
    assert vw.exchanges == 2 and all(c.round == 2 for c in ctxs)


def test_shard_weights_come_from_the_bam_index(synth):
    """`shard_weights` reads the compressed bytes per reference from <bam>.bai (the generator writes one): they add up to the file
    (minus header and EOF block), every reference with records has a weight, and a plan over them is contiguous and complete."""
    import squid_amd
    from squid_amd.dist import bai_ref_weights, plan_shards, shard_weights

    pre = synth("C3", "--records", "200000")
    bam = f"{pre}.bam"
    _, ref_len = squid_amd.read_header(bam)
    w = shard_weights(bam, ref_len)
    assert w == bai_ref_weights(bam + ".bai", len(ref_len)) and len(w) == len(ref_len)
    size = os.path.getsize(bam)
    assert 0.9 * size <= sum(w) <= size + len(ref_len)  # neighbouring references share the BGZF block at their border
    assert sum(1 for x in w if x > 0) >= 20
    for world in (2, 4, 8):
        plan = plan_shards(w, world)
        assert plan[0][0] == 0 and plan[-1][1] == len(ref_len) and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        loads = [sum(w[a:b]) for a, b in plan]
        assert max(loads) <= 1.6 * sum(w) / world  # chr1 alone is 8 % of hg38: 8 contiguous parts cannot be perfectly even
    # no index, a wrong reference count, or garbage -> the lengths
    assert shard_weights(str(pre) + ".missing.bam", ref_len) == list(ref_len)
    assert bai_ref_weights(bam + ".bai", len(ref_len) + 1) is None

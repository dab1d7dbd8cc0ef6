"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

Two modes:
  * sample-parallel -- independent samples, one per rank, no data-path collective (`assign_samples`);
  * chromosome-sharded -- one sample, rank r holds the concordant records of a contiguous RefID range
    (`plan_shards`), and the stage functions of the library pause for variable-length all-gathers
    (`TorchExchange`; include/squid_hip.h, sq_set_shard).  `VirtualWorld` drives several in-process contexts in
    lockstep instead (tests, and one-GPU boxes).
"""
from __future__ import annotations


def assign_samples(n_samples: int, rank: int, world: int) -> list[int]:
    """Round-robin sample -> rank assignment (independent samples: no exchange step)."""
    return [s for s in range(n_samples) if s % world == rank]


def reduce_timing(elapsed: float, n_units: float, dist=None, device="cpu"):
    """(max elapsed over ranks, sum of units over ranks).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), float(n_units)
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    u = torch.tensor([n_units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), float(u[0])


def bai_ref_weights(bai_path, n_ref: int):
    """Compressed bytes of the BAM per reference, from its `.bai` (SAM spec 5.2): the file range in the metadata pseudo-bin 37450
    where the index has one, else from the smallest chunk start to the largest chunk end of the reference's bins.  These are
    the bytes a shard has to inflate -- a better balance for `plan_shards` than the reference lengths.  None when there is no
    usable index (callers fall back to the lengths)."""
    import struct
    try:
        data = open(bai_path, "rb").read()
    except OSError:
        return None
    try:
        if data[:4] != b"BAI\1":
            return None
        (n,) = struct.unpack_from("<i", data, 4)
        if n != n_ref:
            return None
        at, out = 8, []
        for _ in range(n):
            (n_bin,) = struct.unpack_from("<i", data, at); at += 4
            lo, hi, meta = None, None, None
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", data, at); at += 8
                chunks = struct.unpack_from("<%dQ" % (2 * n_chunk), data, at); at += 16 * n_chunk
                if b == 37450:
                    if n_chunk >= 1:
                        meta = (chunks[0], chunks[1])
                    continue
                for k in range(n_chunk):
                    lo = chunks[2 * k] if lo is None else min(lo, chunks[2 * k])
                    hi = chunks[2 * k + 1] if hi is None else max(hi, chunks[2 * k + 1])
            (n_intv,) = struct.unpack_from("<i", data, at); at += 4 + 8 * n_intv
            beg, end = meta if meta else (lo, hi)
            out.append(0 if beg is None or end is None else max(0, (end >> 16) - (beg >> 16)) + (1 if end > beg else 0))
        return out if sum(out) > 0 else None
    except (struct.error, IndexError):
        return None


def shard_weights(bam_path: str, ref_len: list) -> list:
    """what `plan_shards` should balance for this BAM: compressed bytes per reference when `<bam>.bai` is there, else lengths"""
    w = bai_ref_weights(str(bam_path) + ".bai", len(ref_len))
    return w if w is not None else list(ref_len)


def plan_shards(weights: list, world: int) -> list[tuple]:
    """Contiguous RefID ranges [(first, end), ...], one per rank, in rank order, covering all references, with the
    smallest possible maximum weight (weights = records per reference if known, else reference lengths).
    Ranks beyond the number of references get empty ranges."""
    n = len(weights)
    if world <= 0:
        raise ValueError("world must be positive")
    w = [max(0, int(x)) for x in weights]

    def parts_needed(cap: int) -> int:
        parts, run = 1, 0
        for x in w:
            if x > cap:
                return world + 1
            if run + x > cap:
                parts += 1
                run = 0
            run += x
        return parts

    lo, hi = (max(w) if w else 0), sum(w)
    while lo < hi:  # smallest capacity that fits into `world` contiguous parts
        mid = (lo + hi) // 2
        if parts_needed(mid) <= world:
            hi = mid
        else:
            lo = mid + 1
    cap = lo
    out, first, run = [], 0, 0
    for i, x in enumerate(w):
        # close the current part when the next reference does not fit (keep enough references for the remaining ranks is
        # not required: trailing ranks may stay empty)
        if run + x > cap and i > first:
            out.append((first, i))
            first, run = i, 0
        run += x
    out.append((first, n))
    while len(out) < world:
        out.append((n, n))
    return out[:world] if len(out) == world else _merge_tail(out, world)


def _merge_tail(parts: list, world: int) -> list:
    head = parts[: world - 1]
    return head + [(parts[world - 1][0], parts[-1][1])]


class TorchExchange:
    """bytes -> list[bytes]: variable-length all-gather over a torch.distributed group (sizes first, then padded
    payloads).  With the nccl (= RCCL) backend the payload travels through device tensors."""

    def __init__(self, dist, device="cpu", group=None):
        self.dist, self.device, self.group = dist, device, group
        self.calls = 0
        self.bytes = 0

    def __call__(self, blob: bytes) -> list:
        import torch

        dist = self.dist
        world = dist.get_world_size(self.group)
        n = torch.tensor([len(blob)], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(x[0]) for x in sizes]
        cap = max(max(sizes), 1)
        mine = torch.zeros(cap, dtype=torch.uint8)
        if blob:
            mine[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        mine = mine.to(self.device)
        got = [torch.zeros(cap, dtype=torch.uint8, device=self.device) for _ in range(world)]
        dist.all_gather(got, mine, group=self.group)
        self.calls += 1
        self.bytes += sum(sizes)
        return [bytes(g[:s].cpu().numpy().tobytes()) for g, s in zip(got, sizes)]


def install_native_exchange(ctx, dist, backend: str):
    """Let the library carry out its own exchanges (sq_exchange) over torch.distributed's process group:
    nccl -> an RCCL communicator inside the library (rank 0's id travels through the group's store once), ncclAllGather on device
    buffers over xGMI; anything else (gloo: several ranks on one GPU, CPU-side tests) -> a fixed-size all-gather of host buffers
    through the group, installed as the library's transport (sq_set_allgather)."""
    import squid_amd

    import torch

    world = dist.get_world_size()
    if backend == "nccl":
        # ncclCommInitRank is collective: a rank that cannot even bind librccl must not leave the others waiting inside it.  So the
        # ranks first agree that every one of them can (a dlopen, no collective), and only then join the communicator.
        can = torch.tensor([1 if squid_amd.rccl_available() else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(can, op=dist.ReduceOp.MIN)
        if int(can[0]) == 1:
            box = [squid_amd.rccl_unique_id() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ok = 1
            try:
                ctx.rccl_init(box[0])
            except squid_amd.SquidError as e:  # (then on every rank, normally: the ranks agree below and fall back together)
                print(f"squid_amd: sq_rccl_init failed ({e}); the exchanges go through torch.distributed instead", flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                return
            ctx.rccl_release()  # (a rank that did join gives its communicator back before the fallback transport goes in)
        else:
            print("squid_amd: librccl cannot be loaded on every rank; the exchanges go through torch.distributed instead", flush=True)
    dev = "cuda" if backend == "nccl" else "cpu"

    def allgather(blob: bytes) -> bytes:
        mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        got = [torch.zeros(len(blob), dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(got, mine)
        return b"".join(bytes(g.cpu().numpy().tobytes()) for g in got)

    ctx.set_allgather(allgather)


class VirtualWorld:
    """Drives `world` in-process contexts (one per virtual rank, possibly all on one GPU) in lockstep."""

    def __init__(self, contexts: list):
        self.ctxs = contexts
        self.exchanges = 0
        self.bytes = 0
        # the ranks run one after the other here; a real run overlaps them, so the time it would take is the sum over
        # the phases (between two exchanges) of the slowest rank's time in that phase
        self.projected_s = 0.0
        self.serial_s = 0.0
        self.phases = []  # (step name, slowest rank's seconds) per phase between two exchanges

    def _timed(self, c, step_name: str):
        import time

        t0 = time.perf_counter()
        rc = getattr(c, step_name)()
        return rc, time.perf_counter() - t0

    def _lockstep(self, step_name: str):
        pending = list(self.ctxs)
        while True:
            res = [self._timed(c, step_name) for c in pending]
            rcs = [r[0] for r in res]
            self.projected_s += max(r[1] for r in res)
            self.phases.append((step_name, max(r[1] for r in res)))
            self.serial_s += sum(r[1] for r in res)
            if all(rc == 0 for rc in rcs):
                return
            if not all(rc == pending[0].NEED_EXCHANGE for rc in rcs):
                raise RuntimeError(f"ranks out of step: {rcs}")
            parts = [c.exchange_pack() for c in pending]
            self.exchanges += 1
            self.bytes += sum(len(x) for x in parts)
            for c in pending:
                c.exchange_unpack(parts)

    def build_graph(self):
        self._lockstep("build_graph_step")

    def call_sv(self) -> list:
        self._lockstep("call_sv_step")
        return [c.sv_rows() for c in self.ctxs]

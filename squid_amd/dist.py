"""Multi-GPU plumbing for the sample-parallel mode (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).  The hot path has no data-path collective in this mode: ranks only agree
on the slowest rank's time and on the total number of alignments processed."""
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

"""Multi-GPU plumbing of the hot path: one process per GPU, independent camera streams / BA windows
sharded over ranks, no collective on the data path.  The only cross-rank traffic
is a per-frame result summary gathered to every rank with one small all_gather per step
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def stream_assignment(n_streams, world, rank):
    """Stream s runs on rank s % world: returns this rank's stream ids, ascending."""
    return list(range(rank, n_streams, world))


def weak_streams(per_rank, world, rank):
    """Weak scaling: every rank owns `per_rank` streams; global ids are rank-major."""
    return list(range(rank * per_rank, (rank + 1) * per_rank))


class SummaryGather:
    """Gathers a small int32 summary (rows = this rank's frames) from all ranks each step."""

    def __init__(self, rows, cols, device, world):
        self.world = world
        self.local = torch.zeros((rows, cols), dtype=torch.int32, device=device)
        self.all = torch.zeros((world * rows, cols), dtype=torch.int32, device=device) if world > 1 else self.local

    def step(self):
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.local)
        return self.all


def max_over_ranks(value, device, world):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

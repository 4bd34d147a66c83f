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
    """Gathers a small int32 summary (rows = this rank's frames) from all ranks each step.

    Pipelined by one step: step() launches the all_gather of what the caller just wrote into `local` asynchronously
    (on the backend's own stream, ordered after the caller's stream) and returns the result of the PREVIOUS step's
    gather, so the collective's latency overlaps the next step's kernels instead of sitting on the compute stream.
    `local` / the gathered tensors are double-buffered; flush() returns the newest result."""

    def __init__(self, rows, cols, device, world):
        self.world = world
        self._loc = [torch.zeros((rows, cols), dtype=torch.int32, device=device) for _ in range(2)]
        self._all = [torch.zeros((world * rows, cols), dtype=torch.int32, device=device) for _ in range(2)] if world > 1 else self._loc
        self._k = 0
        self._pending = None
        self.local = self._loc[0]

    def step(self):
        """Publishes `local`; returns the fleet summary of the previous step (None on the first call)."""
        prev = self.flush() if (self._pending is not None or self.world == 1 and self._k > 0) else None
        cur = self._k & 1
        if self.world > 1:
            self._pending = (dist.all_gather_into_tensor(self._all[cur], self._loc[cur], async_op=True), self._all[cur])
        else:
            self._pending = (None, self._loc[cur])
        self._k += 1
        self.local = self._loc[self._k & 1]
        return prev

    def flush(self):
        """Waits for the gather in flight (stream-ordered on GPU backends) and returns its result."""
        if self._pending is None:
            return None
        work, out = self._pending
        if work is not None:
            work.wait()
        self._pending = None
        return out


def max_over_ranks(value, device, world):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

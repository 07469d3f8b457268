"""Data-parallel plumbing (one process per GPU, torch.distributed: backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  The path shards on B (independent clips,
SURVEY 8(e)); the only data-path exchange is the gradient all-reduce of the two flat buffers."""
from __future__ import annotations

import torch
import torch.distributed as dist


def balanced_shard(global_batch: int, rank: int, world: int):
    """Rows of the collated global batch (dataloader.py:140-146: piano rows first, violin rows second)
    owned by `rank`: an equal share of EACH half, so every rank sees both labels (a contiguous split
    would give ranks 0..world/2-1 only label 0 and break the class prototypes, style_encoder.py:244-253)."""
    half = global_batch // 2
    if half % world:
        raise ValueError(f"global batch {global_batch} does not split into balanced halves over {world} ranks")
    per = half // world
    piano = list(range(rank * per, (rank + 1) * per))
    violin = [half + i for i in piano]
    return piano + violin


def allreduce_mean_(flat: torch.Tensor, world: int, scale_fn) -> torch.Tensor:
    """In-place mean over ranks of one flat gradient buffer: one collective per optimiser group
    (124 MB for encoders+decoder).  `scale_fn(t, s)` multiplies t by s in place (a HIP kernel on the GPU)."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        scale_fn(flat, 1.0 / world)
    return flat


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)

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


def global_row_order(global_batch: int, world: int):
    """(rank, local row) of every row of the collated global batch, in global order: the inverse of balanced_shard."""
    per = global_batch // 2 // world
    return [(r, h * per + i) for h in (0, 1) for r in range(world) for i in range(per)]


class GatherRowsFn(torch.autograd.Function):
    """All ranks' (b, ...) rows assembled into the (B, ...) global batch in the single-process row order
    (InfoNCE, HSIC, class prototypes and the margin term depend on the whole batch).
    Every rank then evaluates the same global loss L; with the later gradient MEAN over ranks the right
    local gradient is the SUM over ranks of dL/d(global rows), sliced to this rank's rows."""

    @staticmethod
    def forward(ctx, x, rank, world, group):
        ctx.group = group
        ctx.idx = idx = _my_rows(x.shape[0] * world, rank, world, x.device)
        # a SUM all-reduce of rows scattered into a zero global buffer: a few KB, and (unlike all_gather on
        # device tensors) available on both backends the tests use (nccl = RCCL, gloo).  The row indices are a cached DEVICE
        # tensor (built once, outside any capture): index_copy_ / index_select launch plain kernels, so the gather is
        # capturable into the step's hipGraph (a Python list index would upload its indices on every call).
        out = torch.zeros((x.shape[0] * world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        out.index_copy_(0, idx, x.detach())
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g.index_select(0, ctx.idx), None, None, None          # this rank's rows, ascending local row


_row_idx = {}


def _my_rows(global_batch, rank, world, device):
    """Global rows owned by `rank` (ascending local row) as a cached device index tensor."""
    key = (global_batch, rank, world, str(device))
    t = _row_idx.get(key)
    if t is None:
        order = global_row_order(global_batch, world)
        t = _row_idx[key] = torch.tensor([k for k, (r, _) in enumerate(order) if r == rank], dtype=torch.long, device=device)
    return t


def gather_rows(x: torch.Tensor, rank: int, world: int, group=None, force=False) -> torch.Tensor:
    """force: run the collective path on a one-rank group too (the single-GPU rehearsal of the loss-matched mode)."""
    return x if (world == 1 and not force) else GatherRowsFn.apply(x, rank, world, group)


def global_labels(local_labels: torch.Tensor, world: int) -> torch.Tensor:
    """Labels of the gathered batch.  Every rank holds the same balanced layout (balanced_shard), so the global
    labels are the local halves repeated."""
    b = local_labels.numel()
    lo, hi = local_labels[:b // 2], local_labels[b // 2:]
    return torch.cat([lo.repeat(world), hi.repeat(world)])

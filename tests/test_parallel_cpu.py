"""world_size-2 gloo test of the data-parallel plumbing (sharding + flat gradient mean)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ast_amd import parallel


def test_balanced_shard_covers_batch_and_both_labels():
    for B, world in ((16, 2), (64, 8), (8, 1), (32, 4)):
        seen = []
        for r in range(world):
            rows = parallel.balanced_shard(B, r, world)
            assert len(rows) == B // world
            labels = [0 if i < B // 2 else 1 for i in rows]
            assert labels == [0] * (len(rows) // 2) + [1] * (len(rows) // 2)     # same layout as the global batch
            seen += rows
        assert sorted(seen) == list(range(B))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_mean_(flat, world, lambda t, s: t.mul_(s))
    tmax = parallel.max_over_ranks(float(rank + 1), "cpu")
    q.put((rank, flat.tolist(), tmax))            # plain lists: a tensor travels as a shared-memory handle that dies with the worker
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_mean_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.arange(1000, dtype=torch.float32) * 1.5
    for rank, flat, tmax in res:
        assert torch.allclose(torch.tensor(flat), want)
        assert tmax == 2.0


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, d = 8, 5
    glob = torch.arange(B * d, dtype=torch.float32).view(B, d)
    rows = parallel.balanced_shard(B, rank, world)
    x = glob[rows].clone().requires_grad_(True)
    g = parallel.gather_rows(x, rank, world)
    # every rank evaluates the same global loss; its local gradient must be the SUM over ranks of dL/d(global rows)
    w = torch.linspace(0.5, 2.0, B).view(B, 1) * (rank + 1)          # a rank-dependent loss, to see the sum
    (g * w).sum().backward()
    labels = parallel.global_labels(torch.tensor([0, 0, 1, 1]), world)
    q.put((rank, g.detach().tolist(), x.grad.tolist(), rows, labels.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_autograd_gloo_world2():
    """GatherRowsFn: global row order == single-process batch, backward == all-reduced gradient sliced to local rows."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    glob = torch.arange(40, dtype=torch.float32).view(8, 5)
    wsum = torch.linspace(0.5, 2.0, 8).view(8, 1) * 3.0             # (rank 0: x1) + (rank 1: x2)
    for rank, g, grad, rows, labels in res:
        assert torch.equal(torch.tensor(g), glob)
        assert torch.allclose(torch.tensor(grad), wsum[rows].expand(-1, 5))
        assert labels == [0, 0, 0, 0, 1, 1, 1, 1]
    assert parallel.global_row_order(8, 2) == [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (0, 3), (1, 2), (1, 3)]

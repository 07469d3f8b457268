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


def _replica_check_worker(rank, world, port, q):
    """Trainer._check_replicas_after_first_replay on stand-in optimiser groups (CPU tensors, gloo)."""
    import types
    from ast_amd.train import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def group(scale):
        return types.SimpleNamespace(flat_p=torch.arange(100, dtype=torch.float32) * scale, m=torch.full((100,), float(rank)),
                                     v=torch.full((100,), float(rank)), step=torch.tensor([3 + rank]))
    out = []
    for diverged in (False, True):
        tr = types.SimpleNamespace(rank=rank, world=world, G=group(1.0 + (0.5 * rank if diverged else 0.0)), D=group(2.0),
                                   _graphs={"k": 1}, _dist_in_graph=True, _replicas_checked=False)
        Trainer._check_replicas_after_first_replay(tr)
        out.append((tr._dist_in_graph, len(tr._graphs), tr.G.flat_p.tolist(), tr.G.m.tolist()[0], int(tr.G.step)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_check_keeps_identical_ranks_and_resyncs_diverged_ones_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_check_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = (torch.arange(100, dtype=torch.float32)).tolist()
    for rank in (0, 1):
        same, div = res[rank]
        assert same[0] is True and same[1] == 1 and same[2] == base and same[3] == float(rank)      # untouched
        assert div[0] is False and div[1] == 0 and div[2] == base and div[3] == 0.0 and div[4] == 3   # rank 0's state everywhere, graphs dropped

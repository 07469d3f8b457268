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
    q.put((rank, flat.clone(), tmax))
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
        assert torch.allclose(flat, want)
        assert tmax == 2.0

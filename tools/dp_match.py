#!/usr/bin/env python3
"""Loss-matched data parallelism check on ONE GPU: a 2-rank run (gloo, both ranks on cuda:0, sync-BN + gathered
batch-coupled losses) against the single-process step on the same global batch.  Prints one JSON line.
  python tools/dp_match.py [--batch 8] [--sections 1] [--matched 1]"""
import argparse, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run_step(rank, world, args, out_path):
    import ast_amd
    from ast_amd import train
    from ast_amd.parallel import balanced_shard
    ast_amd.set_compute_dtype(torch.float32)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(args.port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x, labels = train.synthetic_batch(args.batch, args.sections, "cuda:0", seed=5)
    if world > 1:
        rows = balanced_shard(args.batch, rank, world)
        x, labels = x[rows].contiguous(), labels[rows]
    cfg = train.TrainConfig(use_graph=False, multi_stream=False, dropout=False, keep_grads=True, loss_matched=bool(args.matched))
    tr = train.Trainer(cfg, rank=rank, world=world, seed=3)
    out = {k: float(v) for k, v in tr.step(x, labels).items()}
    if world > 1:                      # per-row-mean terms: the global value is the mean over ranks
        for k in ("rec", "adv_g", "adv_d", "total"):
            t = torch.tensor([out[k]], dtype=torch.float64); dist.all_reduce(t); out[k] = float(t) / world
    g = tr.last_grad_g
    if rank == 0:
        torch.save({"losses": out, "grad": g[::251].cpu(), "gnorm": float(g.norm()),
                    "bn_rm": tr.style.cnn.net[0].bn1.running_mean.cpu(), "psum": float(tr.G.flat_p.double().abs().sum())}, out_path)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8); ap.add_argument("--sections", type=int, default=1)
    ap.add_argument("--matched", type=int, default=1); ap.add_argument("--port", type=int, default=29533)
    args = ap.parse_args()
    d = tempfile.mkdtemp()
    a, b = os.path.join(d, "single.pt"), os.path.join(d, "dp.pt")
    mp.spawn(run_step, args=(1, args, a), nprocs=1, join=True)
    mp.spawn(run_step, args=(2, args, b), nprocs=2, join=True)
    A, B = torch.load(a), torch.load(b)
    ga, gb = A["grad"].double(), B["grad"].double()
    res = {"matched": bool(args.matched), "losses_single": A["losses"], "losses_dp": B["losses"],
           "grad_rel_err": float((ga - gb).norm() / ga.norm()), "grad_cos": float((ga * gb).sum() / (ga.norm() * gb.norm())),
           "gnorm_single": A["gnorm"], "gnorm_dp": B["gnorm"],
           "bn_running_mean_err": float((A["bn_rm"] - B["bn_rm"]).abs().max()),
           "param_abs_sum_rel": abs(A["psum"] - B["psum"]) / A["psum"]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

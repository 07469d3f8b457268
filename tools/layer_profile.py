#!/usr/bin/env python3
"""Per-launch GEMM table of one eager train step (device events around every igemm / wgrad launch)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import config, ops, train
config.wgrad_defer = False          # time every weight gradient where its layer's backward node launches it
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
ast_amd.set_compute_dtype(dt)
tr = train.Trainer(train.TrainConfig(use_graph=False))
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for _ in range(2): tr.step(x, labels)
torch.cuda.synchronize()
recs = []
orig_ig, orig_wg = ops._igemm, ops._wgrad
def ig(src, wgt, bias, dst, g, flags=0, stats=None, bn=None, per_image=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_ig(src, wgt, bias, dst, g, flags, stats, bn, per_image); e1.record()
    M = g.N*g.Hm*g.Wm
    recs.append(("igemm " + ops._igemm_config(g, ops.dcode(src.dtype)) + (" f32" if src.dtype == torch.float32 else ""), M, g.Cd, g.ntaps*g.Cs, g.ntaps, ops._gemm_cost(g, src.element_size()), e0, e1))
def wg(dy, src, dwp, g, replicas=1, pw=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_wg(dy, src, dwp, g, replicas, pw); e1.record()
    M = g.N*g.Hm*g.Wm
    recs.append(("wgrad" + (" f32" if src.dtype == torch.float32 else ""), M, g.Cd, g.ntaps*g.Cs, g.ntaps, ops._gemm_cost(g, src.element_size(), True), e0, e1))
ops._igemm, ops._wgrad = ig, wg
tr.step(x, labels)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, M, Cd, K, nt, (fl, by), e0, e1 in recs:
    a = agg.setdefault((name, M, Cd, K, nt), [0, 0.0, fl, by])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(a[1] for a in agg.values())
print(f"total GEMM time {tot/1e3:.2f} ms over {len(recs)} launches")
print(f"{'kernel':44s} {'M':>8s} {'Cd':>4s} {'K':>5s} {'taps':>4s} {'n':>3s} {'us/launch':>9s} {'tot_us':>8s} {'TF/s':>7s} {'GB/s':>7s}")
for (name, M, Cd, K, nt), (n, t, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:44s} {M:8d} {Cd:4d} {K:5d} {nt:4d} {n:3d} {t/n:9.1f} {t:8.0f} {fl*n/t/1e6:7.1f} {by*n/t/1e3:7.0f}")

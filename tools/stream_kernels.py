#!/usr/bin/env python3
"""Per-launch table of the HBM-bound kernel families of one eager step (device events around every call):
tools/stream_kernels.py [name substring]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import _lib, train
ast_amd.set_compute_dtype(torch.bfloat16)
tr = train.Trainer(train.TrainConfig(use_graph=False))
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for _ in range(2): tr.step(x, labels)
torch.cuda.synchronize()
_lib.PROFILE_CALLS = []
tr.step(x, labels)
torch.cuda.synchronize()
calls, _lib.PROFILE_CALLS = _lib.PROFILE_CALLS, None
sel = sys.argv[1] if len(sys.argv) > 1 else ""
agg = collections.OrderedDict()
for name, nbytes, e0, e1 in calls:
    if sel in name:
        a = agg.setdefault((name, int(nbytes)), [0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
print(f"{'kernel':28s} {'MB':>8s} {'n':>3s} {'us/launch':>10s} {'GB/s':>8s}")
for (name, nb), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} {nb/1e6:8.2f} {n:3d} {t/n:10.1f} {nb*n/t/1e3:8.0f}")

#!/usr/bin/env python3
"""Call sites of the ATen ops that still launch kernels in one eager train step (TorchDispatchMode + traceback)."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import train
from torch.utils._python_dispatch import TorchDispatchMode
ast_amd.set_compute_dtype(torch.bfloat16)
tr = train.Trainer(train.TrainConfig(use_graph=False, multi_stream=False))
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for _ in range(2): tr.step(x, labels)
torch.cuda.synchronize()
SKIP = ("view", "reshape", "_unsafe_view", "as_strided", "detach", "alias", "expand", "permute", "transpose", "t.", "select.", "slice.",
        "unsqueeze", "squeeze", "empty", "_local_scalar", "unbind", "split", "narrow", "lift_fresh", "is_pinned", "_to_copy", "record_stream")
cnt = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if ("ast_amd" in f.filename or f.filename.endswith("train.py")) and "aten_sites" not in f.filename]
            site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:][::-1]) or "(engine)"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            cnt[(name.replace("aten.", ""), site, shp if len(shp) < 5 else shp)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    tr.step(x, labels)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, s, shp), c in cnt.items(): tot[n] += c
print("totals:", dict(tot.most_common()))
for (n, s, shp), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0][0])):
    print(f"{c:4d} {n:28s} {str(shp):28s} {s}")

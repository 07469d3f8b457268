#!/usr/bin/env python3
"""Which ATen ops (plumbing) still launch kernels in one eager train step, and from where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import train
from torch.profiler import profile, ProfilerActivity
ast_amd.set_compute_dtype(torch.bfloat16)
tr = train.Trainer(train.TrainConfig(use_graph=False, multi_stream=False))
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for _ in range(2): tr.step(x, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x, labels)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
agg = {}
for e in rows:
    a = agg.setdefault(e.key, [0, 0.0]); a[0] += e.count; a[1] += e.device_time_total
print("op, calls, device_us")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"{k:28s} {c:5d} {t:9.1f}")
print("---- top call sites")
for e in rows[:28]:
    st = [s for s in e.stack if "ast_amd" in s or "train.py" in s][:2]
    print(f"{e.key:22s} {e.count:4d} {e.device_time_total:8.1f}us  {' <- '.join(s.split('/')[-1][:60] for s in st)}")

#!/usr/bin/env python3
"""Host-side cost of one replayed step (hipGraphLaunch of ~860 nodes + the input copy) against its GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import train
ast_amd.set_compute_dtype(torch.bfloat16)
tr = train.Trainer(train.TrainConfig())
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for _ in range(5): tr.step(x, labels)
torch.cuda.synchronize()
for label, n in (("enqueue only", 20), ("enqueue only", 20)):
    t0 = time.perf_counter()
    for _ in range(n): tr.step(x, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, until GPU idle {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)

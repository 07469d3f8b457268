#!/usr/bin/env python3
"""One-GPU timing of the three-graph (data-parallel) form of the step against the single-graph form."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import train
ast_amd.set_compute_dtype(torch.bfloat16)
x, labels = train.synthetic_batch(8, 2, "cuda:0")
for seg in (False, True):
    tr = train.Trainer(train.TrainConfig(segmented=seg))
    for _ in range(5): tr.step(x, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tr.step(x, labels)
    torch.cuda.synchronize()
    print(f"segmented={seg}: {(time.perf_counter() - t0) * 10:.3f} ms/step", flush=True)
    del tr; torch.cuda.empty_cache()

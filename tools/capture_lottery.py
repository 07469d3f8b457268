#!/usr/bin/env python3
"""Does the replayed step's time depend on the capture?  One process, one trainer: capture the step K times (dropping the
previous graph), time 20 replays of each.  tools/capture_lottery.py [K=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
import ast_amd
from ast_amd import train
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda:0"
ast_amd.set_compute_dtype(torch.bfloat16)
tr = train.Trainer(train.TrainConfig(use_graph=True), device=dev)
waves, x, mean, std, labels = train.synthetic_waveform_batch(8, 4.0, dev, seed=1000)
tr.set_frontend(waves, mean, std, torch.zeros(2, 84, device=dev), torch.full((2, 84), 0.25, device=dev))
for k in range(K):
    tr._graphs.clear()
    for _ in range(5):
        tr.step(x, labels)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(x, labels)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"capture {k}: " + " ".join(f"{t:.3f}" for t in ts) + " ms/step", flush=True)

#!/usr/bin/env python3
"""Run-to-run spread of the f32 step's loss scalars (atomic-order noise): eager twice, graph twice."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch, ast_amd
from ast_amd import train, utilityFunctions as U
ast_amd.set_compute_dtype(torch.float32)
waves, x, mean, std, labels = train.synthetic_waveform_batch(2, 4.0, "cuda:0", seed=9)
x_ref = x.clone(); U.stft_sections(waves, mean, std, n_sections=2, F_total=597, out=x_ref)
runs = []
for i in range(6):
    t = train.Trainer(train.TrainConfig(use_graph=(i % 2 == 1), dropout=False), seed=3)
    if i % 2 == 1:
        t.set_frontend(waves, mean, std); r = t.step(x.clone(), labels)
    else:
        r = t.step(x_ref, labels)
    runs.append({k: float(v) for k, v in r.items()})
for k in runs[0]:
    vals = [r[k] for r in runs]
    spread = (max(vals) - min(vals)) / (abs(vals[0]) + 1e-12)
    print(f"{k:28s} {vals[0]: .6e} spread {spread:.2e}  " + " ".join(f"{v:.6e}" for v in vals))

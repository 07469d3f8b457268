#!/usr/bin/env python3
"""Run-to-run spread of the f32 whole-model gradient error against the on-box oracle (tests/test_gpu_models.py's
check): one oracle step, several HIP steps from fresh seeded models."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_models as T
import ast_amd
from oracle import seeded_params as sp
ast_amd.set_compute_dtype(torch.float32)
B, S = 2, 2
o = T.oracle_step(B, S)
x, labels = sp.seeded_input(B, S).to("cuda"), sp.balanced_labels(B)
for it in range(6):
    ms = T.build_models()
    T.hip_step(ms, x, labels)
    torch.cuda.synchronize()
    res = []
    for tag in ("style", "content", "decoder"):
        num = den = 0.0
        worst = 0.0
        for k, p in ms[tag].named_parameters():
            ref = o["sds"][tag][k].grad
            if ref is None or p.grad is None: continue
            num += float((p.grad.double().cpu() - ref.double()).pow(2).sum()); den += float(ref.double().pow(2).sum())
            if float(ref.norm()) >= 1e-3:
                worst = max(worst, T.rel_l2(p.grad, ref))
        res.append(f"{tag} {math.sqrt(num / den):.2e} (worst parameter {worst:.1e})")
    print(it, " ".join(res), flush=True)

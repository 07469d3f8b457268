#!/usr/bin/env python3
"""Device time of the gradient-norm pass (ast_sumsq over the 31 M-parameter flat gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd"))
import torch
from ast_amd._lib import lib, check, ptr, stream
x = torch.randn(31071408, device="cuda")
out = torch.zeros(1, device="cuda")
for _ in range(3):
    check(lib().ast_sumsq(ptr(x), x.numel(), ptr(out), stream()), "ast_sumsq")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out.zero_(); e0.record()
for _ in range(50):
    check(lib().ast_sumsq(ptr(x), x.numel(), ptr(out), stream()), "ast_sumsq")
e1.record(); torch.cuda.synchronize()
ref = float((x.double() ** 2).sum())
print(f"sumsq: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, rel err {abs(float(out) / 50 - ref) / ref:.1e}")

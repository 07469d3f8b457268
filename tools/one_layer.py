#!/usr/bin/env python3
"""Run one conv layer's GEMM a few times (for rocprofv3 --pmc): tools/one_layer.py b0c2 [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
SH = {"b5c2": (16, 5, 10, 512, 512, 3, 1), "b4c2": (16, 9, 19, 512, 512, 3, 1), "b3c2": (16, 18, 38, 256, 256, 3, 1),
      "b2c2": (16, 36, 75, 128, 128, 3, 1), "b1c2": (16, 72, 150, 64, 64, 3, 1), "b0c2": (16, 144, 299, 32, 32, 3, 1),
      "dec0": (16, 287, 513, 8, 16, 3, 1)}
name = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, H, W, Cs, Cd, k, st = SH[name]
dt = torch.bfloat16
g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, st, 1)
x = torch.randn(N, H, W, Cs, device="cuda").to(dt); w = torch.randn(Cd, k * k, Cs, device="cuda").to(dt)
y = torch.empty(N, Ho, Wo, Cd, device="cuda", dtype=dt)
ws = torch.zeros(max(1, N * Ho * Wo * Cd), device="cuda")
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "wgrad":          # the weight gradient of the same layer: dW[Cd][taps][Cs] += dy^T . gather(x)
    dy = torch.randn(N, Ho, Wo, Cd, device="cuda").to(dt)
    dw = torch.zeros(Cd, k * k, Cs, device="cuda")
    for _ in range(iters):
        check(lib().ast_wgrad(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), stream()))
else:
    for _ in range(iters):
        check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 4, ptr(ws), ws.numel(), stream()))
torch.cuda.synchronize()

#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
dt = torch.bfloat16
SH = [("b5c2", 16, 5, 10, 512, 512, 3, 1), ("b3c2", 16, 18, 38, 256, 256, 3, 1), ("b2c2", 16, 36, 75, 128, 128, 3, 1), ("b1c2", 16, 72, 150, 64, 64, 3, 1),
      ("b1c1", 16, 144, 299, 32, 64, 3, 2), ("b0c2", 16, 144, 299, 32, 32, 3, 1), ("b0c1", 16, 287, 597, 8, 32, 3, 2), ("dec0", 16, 287, 513, 8, 16, 3, 1)]
for name, N, H, W, Cs, Cd, k, st in SH:
    g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, st, 1)
    x = torch.randn(N, H, W, Cs, device="cuda").to(dt); dy = torch.randn(N, Ho, Wo, Cd, device="cuda").to(dt)
    dw = torch.zeros(Cd, k * k, Cs, device="cuda")
    for _ in range(3): check(lib().ast_wgrad(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): check(lib().ast_wgrad(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), stream()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = (x.numel() + dy.numel()) * 2
    print(f"{name} P={N*Ho*Wo} Cd={Cd} cols={k*k*Cs}: {us:7.1f} us  {by/us/1e3:7.0f} GB/s (hbm ideal {by/6e6:.1f}us)", flush=True)

#!/usr/bin/env python3
"""Isolated timing of the token-path skinny GEMM shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd._lib import lib, check, ptr, stream
for (M, N, K) in ((24, 256, 1024), (24, 1024, 256), (24, 768, 256), (24, 256, 256), (16, 256, 1024), (40, 256, 1024), (24, 512, 256)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    def run():
        check(lib().ast_skinny_gemm(ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, K, N, 0, stream()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    ref = x @ w.t() + b
    err = float((y - ref).abs().max() / ref.abs().max())
    print(f"M={M:3d} N={N:5d} K={K:5d}: {e0.elapsed_time(e1) / 200 * 1e3:6.2f} us/launch (back-to-back)  err {err:.1e}", flush=True)

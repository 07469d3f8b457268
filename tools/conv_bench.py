#!/usr/bin/env python3
"""Time the forward GEMM of the encoder's stride-1 3x3 layers (and optionally their weight gradients) in isolation:
tools/conv_bench.py [layers=b0c2,b1c2,b2c2,b3c2] [iters=30] [wgrad]   -- one line per layer: us, TFLOP/s.
NOSTATS=1: plain stores; PERIMG=1: per-image statistics (flags 8|64); default: fused BatchNorm statistics (flags 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
SH = {"b5c2": (16, 5, 10, 512, 512, 3, 1), "b4c2": (16, 9, 19, 512, 512, 3, 1), "b3c2": (16, 18, 38, 256, 256, 3, 1),
      "b2c2": (16, 36, 75, 128, 128, 3, 1), "b1c2": (16, 72, 150, 64, 64, 3, 1), "b0c2": (16, 144, 299, 32, 32, 3, 1),
      "b1c1": (16, 144, 299, 32, 64, 3, 2), "b2c1": (16, 72, 150, 64, 128, 3, 2),
      "dec0": (16, 287, 513, 8, 16, 3, 1), "b0c1": (16, 287, 597, 8, 32, 3, 2), "b0ds": (16, 287, 597, 8, 32, 1, 2),
      "dec1": (16, 287, 513, 16, 32, 3, 2)}
layers = (sys.argv[1] if len(sys.argv) > 1 else "b0c2,b1c2,b2c2,b3c2").split(",")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wgrad = len(sys.argv) > 3 and sys.argv[3] == "wgrad"
dt = torch.bfloat16
for name in layers:
    N, H, W, Cs, Cd, k, st = SH[name]
    g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, st, 1 if k == 3 else 0)
    x = torch.randn(N, H, W, Cs, device="cuda").to(dt); w = (0.05 * torch.randn(Cd, k * k, Cs, device="cuda")).to(dt)
    y = torch.empty(N, Ho, Wo, Cd, device="cuda", dtype=dt)
    stats = torch.zeros(64 * Cd * 2, device="cuda")
    wsz = torch.zeros(N * Ho * Wo * Cd, device="cuda")
    rep = int(os.environ.get("WGRAD_REP", "1"))                 # gradient replicas (ast_wgrad_rep)
    dy = torch.randn(N, Ho, Wo, Cd, device="cuda").to(dt); dw = torch.zeros(rep, Cd, k * k, Cs, device="cuda")
    nsl = int(os.environ.get("WGRAD_SLABS", "0"))             # slab flush (ast_wgrad_slab): plain stores into one copy per pixel slice
    if nsl:
        import ctypes
        slabs = torch.zeros(nsl, Cd, k * k, Cs, device="cuda"); sl_out = ctypes.c_int(0)
    def run():
        if wgrad:
            if nsl:
                check(lib().ast_wgrad_slab(ptr(dy), ptr(x), ptr(slabs), g, dcode(dt), nsl, ctypes.byref(sl_out), stream()))
            elif rep > 1:
                check(lib().ast_wgrad_rep(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), rep, stream()))
            else:
                check(lib().ast_wgrad(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), stream()))
        else:
            if os.environ.get("NOSTATS"):
                check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 4, ptr(wsz), wsz.numel(), stream()))
            elif os.environ.get("PERIMG"):                       # per-image (InstanceNorm) sums: flags bit 6, [N][Cd][2] table
                check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 8 | 64, ptr(stats), stats.numel(), stream()))
            else:
                check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 8, ptr(stats), stats.numel(), stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * N * Ho * Wo * Cd * k * k * Cs
    nbytes = (N * Ho * Wo * Cd + N * H * W * Cs) * 2
    print(f"{name} {'wgrad' if wgrad else 'fwd+stats'} {ops._igemm_config(g, dcode(dt)) if not wgrad else ''}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s  {nbytes / us / 1e3:6.0f} GB/s", flush=True)

#!/usr/bin/env python3
"""Sweep tile plans (AST_IGEMM_FORCE=bm,bn,kch,nsplit) for the conv shapes of the B=8,S=2 step."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
dt = torch.bfloat16
SHAPES = [  # (N,H,W,Cs,Cd,k,stride) forward convs
    ("b5c2 800x512x4608", 16, 5, 10, 512, 512, 3, 1),
    ("b4c2 2736x512x4608", 16, 9, 19, 512, 512, 3, 1),
    ("b3c2 10944x256x2304", 16, 18, 38, 256, 256, 3, 1),
    ("b2c2 43200x128x1152", 16, 36, 75, 128, 128, 3, 1),
    ("b1c2 172800x64x576", 16, 72, 150, 64, 64, 3, 1),
    ("b0c2 688896x32x288", 16, 144, 299, 32, 32, 3, 1),
    ("dec0 2355696x16x72", 16, 287, 513, 8, 16, 3, 1),
    ("b4ds 2736x512x256(1x1s2)", 16, 18, 38, 256, 512, 1, 2),
]
PLANS = {  # (bm, bn, kch, grid split-K, in-workgroup K groups)
    "deep": [(64, 64, k, s, kg) for k in (8, 4) for s in (1, 2, 4) for kg in (1, 4)] + [(64, 128, 8, s, 1) for s in (1, 4)] + [(128, 64, 8, 1, 1)],
    "mid": [(bm, bn, k, 1, 1) for (bm, bn) in ((128, 128), (128, 64), (64, 64), (64, 128)) for k in (4, 8)] + [(64, 64, k, 1, 4) for k in (4, 8)],
    "wide": [(bm, bn, 4, 1, 1) for (bm, bn) in ((256, 32), (128, 32), (64, 32), (256, 16), (128, 16), (64, 16), (64, 64), (128, 64))] + [(64, 64, 4, 1, 4)],
}
def run(name, N, H, W, Cs, Cd, k, stride, plans):
    pad = 1 if k == 3 else 0
    g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, stride, pad)
    x = torch.randn(N, H, W, Cs, device="cuda").to(dt)
    w = torch.randn(Cd, k * k, Cs, device="cuda").to(dt)
    y = torch.empty(N, Ho, Wo, Cd, device="cuda", dtype=dt)
    M = N * Ho * Wo
    ws = torch.zeros(4 * M * Cd, device="cuda")
    flops = 2.0 * M * Cd * k * k * Cs
    nbytes = (x.numel() + y.numel() + w.numel()) * 2
    out = []
    os.environ["AST_IGEMM_FORCE"] = "64,64,4,1,1"
    yref = torch.empty_like(y)
    check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(yref), g, dcode(dt), 0, ptr(ws), ws.numel(), stream()))
    for pl in plans:
        if pl[1] > max(Cd, 16) * 2: continue
        os.environ["AST_IGEMM_FORCE"] = ",".join(map(str, pl))
        try:
            for _ in range(3):
                check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 0, ptr(ws), ws.numel(), stream()))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 0, ptr(ws), ws.numel(), stream()))
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            err = (y.float() - yref.float()).abs().max().item() / yref.float().abs().max().item()
            out.append((us if err < 2e-2 else float("nan"), pl))
        except Exception as ex:
            out.append((float("inf"), pl))
    out.sort()
    print(f"{name:28s} ideal hbm {nbytes/6e6:6.1f}us mfma {flops/2.5e9:6.1f}us | " + "  ".join(f"{pl}:{us:.1f}" for us, pl in out[:6]), flush=True)
for sh in SHAPES:
    name = sh[0]
    M = sh[1] * ((sh[2] + 2 * (1 if sh[6] == 3 else 0) - sh[6]) // sh[7] + 1) * ((sh[3] + 2 * (1 if sh[6] == 3 else 0) - sh[6]) // sh[7] + 1)
    kind = "deep" if M < 20000 else ("mid" if sh[5] >= 64 else "wide")
    run(*sh, PLANS[kind])

#!/usr/bin/env python3
"""Per-op cost inside a token program: programs of 24 identical ops (each followed by a grid barrier), time per op."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from ast_amd import tokprog as T
from ast_amd._lib import lib
dev = torch.device("cuda:0")
rows, d, F = 24, 256, 1024
x = torch.randn(rows, 1024, device=dev); w = torch.randn(1024, 1024, device=dev) * 0.05; b = torch.randn(1024, device=dev)
y = torch.empty(rows, 1024, device=dev); g = torch.ones(d, device=dev); st = torch.empty(2, rows, device=dev)
y2 = torch.empty(rows, 1024, device=dev); probs = torch.empty(8 * 8 * 16 * 16, device=dev)
def timeit(name, op, n=24):
    ops = [op] * n
    for _ in range(3): T.run(ops, dev, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): T.run(ops, dev, 0)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) * 1000 / 10
    extra = ""
    if hasattr(lib(), "ast_debug_read_tok_stamps") and n > 4:
        buf = (ctypes.c_ulonglong * (32 * 64 * 3))()
        f = lib().ast_debug_read_tok_stamps; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
        assert f(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(32, 64, 3).astype(np.int64)[:T.G_WORKGROUPS, 2:n - 2]
        comp = (a[:, :, 1] - a[:, :, 0]) / 100.0; bar = comp * 0
        clk = np.median((a[0, 1:, 2] - a[0, :-1, 2]) / ((a[0, 1:, 0] - a[0, :-1, 0]) / 100.0)) / 1e3
        extra = f"   compute: wg0 {np.median(comp[0]):.2f} max-over-wgs {np.median(comp.max(0)):.2f} min {np.median(comp.min(0)):.2f} us; shader clock {clk:.2f} GHz"
    print(f"{name:44s} {tot / n:6.2f} us per op   ({tot:.0f} us per launch of {n}){extra}", flush=True)
for G in (16, 32):
    T.G_WORKGROUPS = G
    print("G =", G)
    timeit("empty launch (1 op: 1-row LN)", T._op(T.ADLN_FWD, 0, (1, d), (None, x, g, g), (None, None, y, st[0], st[1]), eps=1e-5), n=1)
    timeit("ADLN_FWD rows=24", T._op(T.ADLN_FWD, 0, (rows, d), (x, y2, g, g), (None, y, y, st[0], st[1]), eps=1e-5))
    timeit("GEMM 24 x 768 x 256 (qkv)", T.gemm(x, w, b, y, rows, 768, 256, 256, 256, 768))
    timeit("GEMM 24 x 256 x 256 (out proj, K split)", T.gemm(x, w, b, y, rows, 256, 256, 256, 256, 256))
    timeit("GEMM 24 x 1024 x 256 (ffn1)", T.gemm(x, w, b, y, rows, 1024, 256, 256, 256, 1024, relu=True))
    timeit("GEMM 24 x 256 x 1024 (ffn2, K split)", T.gemm(x, w, b, y, rows, 256, 1024, 1024, 1024, 256))
    timeit("ATTN_FWD B=8 H=8 L=3", T._op(T.ATTN_FWD, 0, (8, 8, 3, 3, 32, 768, 768, 256), (x, (x, 256), (x, 512)), (y2, probs)))
T.check_status()

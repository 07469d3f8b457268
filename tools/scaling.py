#!/usr/bin/env python3
"""Time of one igemm launch against K (Cs) and M (image height): separates fixed cost from the K loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
dt = torch.bfloat16
def t(N, H, W, Cs, Cd, plan, k=3):
    g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, 1, k // 2)
    x = torch.randn(N, H, W, Cs, device="cuda").to(dt); w = torch.randn(Cd, k * k, Cs, device="cuda").to(dt)
    y = torch.empty(N, Ho, Wo, Cd, device="cuda", dtype=dt); ws = torch.zeros(4 * N * Ho * Wo * Cd, device="cuda")
    os.environ["AST_IGEMM_FORCE"] = plan
    for _ in range(3): check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 0, ptr(ws), ws.numel(), stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 0, ptr(ws), ws.numel(), stream()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3
for plan in ("64,64,8,1,1,0", "128,128,8,1,1,1"):
    print("plan", plan)
    print("  K sweep (M=2736, Cd=512): " + "  ".join(f"Cs={cs}:{t(16, 9, 19, cs, 512, plan):.1f}us" for cs in (64, 128, 256, 512, 1024)), flush=True)
    print("  1x1 K sweep (M=2736, Cd=512): " + "  ".join(f"Cs={cs}:{t(16, 9, 19, cs, 512, plan, 1):.1f}us" for cs in (64, 512, 2048, 4096)), flush=True)
    print("  M sweep (Cs=Cd=512): " + "  ".join(f"M={16*h*19}:{t(16, h, 19, 512, 512, plan):.1f}us" for h in (2, 4, 9, 18, 36)), flush=True)
    print("  M sweep (Cs=Cd=128): " + "  ".join(f"M={16*h*75}:{t(16, h, 75, 128, 128, plan):.1f}us" for h in (2, 9, 36, 72, 144)), flush=True)

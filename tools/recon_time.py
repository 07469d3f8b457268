#!/usr/bin/env python3
"""Device time of the one-pass reconstruction loss (compute_comprehensive_loss fwd + gradient) at B=8, S=2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd"))
import torch, ast_amd
out = torch.randn(8, 2, 2, 287, 513, device="cuda", requires_grad=True)
x = torch.randn(8, 2, 2, 287, 597, device="cuda")
tgt = x[..., :513]
for _ in range(3):
    ast_amd.compute_comprehensive_loss(out, tgt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ast_amd.compute_comprehensive_loss(out, tgt)
e1.record(); torch.cuda.synchronize()
print(f"recon loss: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (kernel + memset + 2 ATen glue launches)")

#!/usr/bin/env python3
"""Time the device CQT front end (get_CQT + z-score + sectioning) on a batch of 4 s clips; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd"))
import torch
from ast_amd import cqt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = 0.07 * torch.randn(B, 88200, device="cuda")
x = torch.zeros(B, 2, 2, 287, 597, device="cuda")
for _ in range(3):
    cqt.cqt_sections(w, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    cqt.cqt_sections(w, x)
e1.record(); torch.cuda.synchronize()
print(f"cqt_sections B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per batch (eager, 14 launches)")

#!/usr/bin/env python3
"""Phase timeline of wgrad_kernel from in-kernel stamps (debug build: make -C audio-style-transfer_amd/csrc stamps).
tools/wgrad_stamps.py [layer=b1c2]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AST_HIP_LIB", os.path.join(ROOT, "audio-style-transfer_amd", "ast_amd", "libast_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
SH = {"b4c2": (16, 9, 19, 512, 512, 3, 1), "b3c2": (16, 18, 38, 256, 256, 3, 1), "b2c2": (16, 36, 75, 128, 128, 3, 1), "b1c2": (16, 72, 150, 64, 64, 3, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "b1c2"
N, H, W, Cs, Cd, k, st = SH[name]
dt = torch.bfloat16
g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, st, 1)
x = torch.randn(N, H, W, Cs, device="cuda").to(dt)
dy = torch.randn(N, Ho, Wo, Cd, device="cuda").to(dt); dw = torch.zeros(Cd, k * k, Cs, device="cuda")
for _ in range(3):
    check(lib().ast_wgrad(ptr(dy), ptr(x), ptr(dw), g, dcode(dt), stream()))
torch.cuda.synchronize()
n = 4096
buf = (ctypes.c_ulonglong * (n * 8))()
f = lib().ast_debug_read_wg_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
assert f(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
a = a[a[:, 0] != 0]
print(f"{name}: {len(a)} workgroups stamped")
for i, nm in enumerate(["prologue (tap table, loader descriptors, first loads)", "K loop over the pixel slice", "pixel-group reduction through LDS", "atomic flush of the tile"]):
    d = a[:, i + 1] - a[:, i]
    print(f"  {nm:55s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} cycles")
ph = (ctypes.c_longlong * (n * 4))()
f2 = lib().ast_debug_read_wg_phases; f2.argtypes = [ctypes.c_void_p, ctypes.c_int]; f2.restype = ctypes.c_int
assert f2(ph, n) == 0
pa = np.frombuffer(ph, dtype=np.int64).reshape(n, 4)[:len(a)]
trips = np.maximum(pa[:, 3], 1)
sub = np.stack([a[:, 5] >> 32, a[:, 5] & 0xffffffff, pa[:, 0], pa[:, 1], pa[:, 2]], axis=1) / trips[:, None]
rows = os.environ.get("AST_WGRAD_ROWS", "1") != "0" and Cs % 64 == 0 and Cd % 64 == 0 and st == 1      # the line-staged kernel's sub-phases
names = (["wait for the tile's DMAs", "barrier", "reads of k-step 0 (returned) + DMA issue", "reads of k-step 1 (returned) + MFMAs 0 issued", "MFMAs 1 issued"] if rows else
         ["LDS store + wait for the loads", "barrier", "issue next loads", "LDS reads + MFMA", "barrier"])
print(f"  K loop per trip ({int(np.median(trips))} trips, thread 0): " + ", ".join(f"{nm} {np.median(sub[:, i]):.0f}" for i, nm in enumerate(names)) + " cycles")
life = a[:, 4] - a[:, 0]
us = (a[:, 6] - a[:, 7]) / 100.0
print(f"  workgroup lifetime median {np.median(life):.0f} cycles = {np.median(us):.2f} us (shader clock {np.median(life / us) / 1e3:.2f} GHz); last workgroup ends at {(a[:, 6].max() - a[:, 7].min()) / 100.0:.1f} us")
t0 = a[:, 7] - a[:, 7].min()
print("  workgroup starts per microsecond:", np.bincount((t0 // 100).astype(int)).tolist()[:20])

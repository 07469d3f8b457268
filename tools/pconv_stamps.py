#!/usr/bin/env python3
"""Phase timeline of the patch-staged conv kernel from in-kernel shader-clock stamps (debug build: `make -C
audio-style-transfer_amd/csrc stamps`; run with AST_HIP_LIB=.../libast_hip_stamps.so).
tools/pconv_stamps.py [layer=b1c2]   -- per-phase cycles of thread 0 of every workgroup (median / p90), and how many
workgroups were resident over time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AST_HIP_LIB", os.path.join(ROOT, "audio-style-transfer_amd", "ast_amd", "libast_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from ast_amd import ops
from ast_amd._lib import lib, check, ptr, stream, dcode
SH = {"b3c2": (16, 18, 38, 256, 256, 3, 1), "b2c2": (16, 36, 75, 128, 128, 3, 1), "b1c2": (16, 72, 150, 64, 64, 3, 1),
      "b0c2": (16, 144, 299, 32, 32, 3, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "b1c2"
N, H, W, Cs, Cd, k, st = SH[name]
dt = torch.bfloat16
g, (Ho, Wo) = ops.gather_direct(N, H, W, Cs, Cd, k, st, 1)
x = torch.randn(N, H, W, Cs, device="cuda").to(dt); w = (0.05 * torch.randn(Cd, k * k, Cs, device="cuda")).to(dt)
y = torch.empty(N, Ho, Wo, Cd, device="cuda", dtype=dt); stats = torch.zeros(64 * Cd * 2, device="cuda")
for _ in range(3):
    check(lib().ast_igemm(ptr(x), ptr(w), None, ptr(y), g, dcode(dt), 8, ptr(stats), stats.numel(), stream()))
torch.cuda.synchronize()
raw = lib()
n = 16384
buf = (ctypes.c_ulonglong * (n * 8))()
f = raw.ast_debug_read_stamps; f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
assert f(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
a = a[a[:, 0] != 0]
print(f"{name}: {ops._igemm_config(g, dcode(dt))}, {len(a)} workgroups stamped")
names = ["prologue (index math, descriptors)", "patch + tap-0 weights -> LDS", "tap loop", "epilogue stores", "stat flush"]
for i, nm in enumerate(names):
    d = a[:, i + 1] - a[:, i]
    print(f"  {nm:40s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} cycles")
life = a[:, 5] - a[:, 0]
print(f"  {'workgroup lifetime':40s} median {np.median(life):8.0f}  p10 {np.percentile(life, 10):8.0f}  p90 {np.percentile(life, 90):8.0f} cycles")
t0 = a[:, 7] - a[:, 7].min()                      # wall clock (100 MHz) at workgroup start
us = (a[:, 6] - a[:, 7]) / 100.0                 # the same lifetime on the 100 MHz wall clock
print(f"  lifetime on the wall clock: median {np.median(us):.2f} us -> shader clock ~{np.median(life / us) / 1e3:.2f} GHz; last workgroup ends at {(a[:, 6].max() - a[:, 7].min()) / 100.0:.1f} us")
print("  workgroup starts per microsecond:", np.bincount((t0 // 100).astype(int)).tolist())

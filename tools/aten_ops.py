#!/usr/bin/env python3
"""Which ATen operators (each a kernel launch of its own in the replayed graph) one training step still issues, by Python
call site: an eager step of the benchmarked configuration under the CPU-side torch profiler with stacks.
tools/aten_ops.py [B=8] [S=2]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ast_amd import config
from ast_amd.train import Trainer, TrainConfig, synthetic_waveform_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
config.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda:0")
tr = Trainer(TrainConfig(use_graph=False), device=dev)
waves, x, mean, std, labels = synthetic_waveform_batch(B, 2.0 * S, dev)
tr.set_frontend(waves, mean, std, torch.zeros(2, 84, device=dev), torch.full((2, 84), 0.25, device=dev))
for _ in range(2):
    tr.step(x, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(x, labels)
    torch.cuda.synchronize()
LAUNCHING = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::copy_", "aten::div", "aten::div_",
             "aten::sum", "aten::mean", "aten::sub", "aten::neg", "aten::cat", "aten::index_select", "aten::clone", "aten::_to_copy",
             "aten::where", "aten::exp", "aten::log", "aten::sqrt", "aten::clamp", "aten::clamp_min", "aten::relu", "aten::index", "aten::masked_fill_",
             "aten::stack", "aten::sigmoid", "aten::tanh", "aten::pow", "aten::mm", "aten::bmm", "aten::addmm", "aten::gather", "aten::scatter_",
             "aten::lerp_", "aten::addcmul_", "aten::addcdiv_", "aten::norm", "aten::linalg_vector_norm", "aten::max", "aten::min", "aten::eq", "aten::ne")
sites = collections.Counter()
for ev in prof.events():
    if ev.name not in LAUNCHING:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name in LAUNCHING + ("aten::zeros", "aten::zeros_like", "aten::ones", "aten::full", "aten::to", "aten::contiguous"):
        continue                                   # count the outermost launching op only
    site = "autograd engine (gradient accumulation / no Python frame)"
    for fr in ev.stack or []:
        if "/ast_amd/" in fr or "/oracle/" in fr or "bench.py" in fr:
            site = fr.split("/ast_amd/")[-1] if "/ast_amd/" in fr else fr
            break
    else:
        if ev.stack:
            site = "stack: " + " <- ".join(f.split("/")[-1] for f in ev.stack[:3])
    shp = str([tuple(x) for x in (ev.input_shapes or []) if x][:2])
    sites[(ev.name, site + " " + shp)] += 1
tot = sum(sites.values())
print(f"{tot} launching ATen ops in one eager step (B={B}, S={S})")
for (name, site), n in sites.most_common(60):
    print(f"{n:4d}  {name:18s} {site}")

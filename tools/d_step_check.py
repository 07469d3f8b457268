#!/usr/bin/env python3
"""Diagnostic: the discriminator's first optimiser step (Trainer, f32) against the oracle's -- gradient, sign flips of the
first Adam update, updated parameters, and the generator's adversarial term evaluated behind it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ast_amd
from ast_amd import train
from oracle import seeded_params as sp
from oracle.train_step import OracleTrainer
import test_gpu_bench_config as T

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 2
ot = OracleTrainer()
x, labels = sp.seeded_input(B, S), sp.balanced_labels(B)
p0 = {k: v.detach().clone() for k, v in ot.sds["disc"].items()}
ref = ot.step(x, labels, apply_g=False)
# oracle D gradient = recompute: (p0 - p1) sign pattern; the raw gradient is gone after od.step(), so redo the D phase
ot2 = OracleTrainer()
terms, _ = ot2.forward_losses(x, labels, with_d_step=False)
terms["adv_d"].requires_grad_(False)
from oracle import ast_oracle as O
st, cl, co = ot2.embeddings
d_loss, _ = O.adversarial_loss(ot2.sds["disc"], st, cl, co, labels, True)
d_loss.backward()
gd_ref = {k: v.grad.clone() for k, v in ot2.sds["disc"].items() if v.requires_grad}

tr = T.seeded_trainer(torch.float32, use_graph=False)
grabbed = {}
orig = tr.D.adam
def adam_spy(*a, **k):
    grabbed["g"] = tr.D.flat_g.clone(); grabbed["p_before"] = tr.D.flat_p.clone()
    orig(*a, **k)
    grabbed["p_after"] = tr.D.flat_p.clone()
tr.D.adam = adam_spy
out = {k: float(v) for k, v in tr.step(x.cuda(), labels).items()}
torch.cuda.synchronize()
print("losses hip   ", out)
print("losses oracle", ref)
off = 0
for (name, p) in tr.disc.named_parameters():
    k = p.numel()
    g = grabbed["g"][off:off + k].view(p.shape).cpu(); pa = grabbed["p_after"][off:off + k].view(p.shape).cpu(); pb = grabbed["p_before"][off:off + k].view(p.shape).cpu()
    off += (k + 3) // 4 * 4
    gr = gd_ref[name]
    rel = float((g - gr).norm() / gr.norm())
    flips = int(((g * gr) < 0).sum()); small = int((gr.abs() < 1e-3 * gr.abs().max()).sum()); zeros = int((gr == 0).sum())
    dp_h, dp_o = pa - pb, ot.sds["disc"][name].detach() - p0[name]
    print(f"{name:16s} n={k:6d} grad rel-L2 {rel:.2e}  sign flips {flips:5d}  |g|<1e-3max {small:5d}  exact zeros {zeros:5d}  "
          f"update mismatch (>1e-5) {int(((dp_h - dp_o).abs() > 1e-5).sum()):5d}  mean|dp| hip {float(dp_h.abs().mean()):.2e} oracle {float(dp_o.abs().mean()):.2e}")

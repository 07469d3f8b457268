#!/usr/bin/env python3
"""Which STAGE's bf16 storage rounding produces the embedding error and the margin / HSIC gradient error of the bf16 mode?

CPU only (the oracle against itself): the oracle with bf16 storage emulated (oracle.Cfg(act_dtype=bf16), which reproduces the
HIP bf16 mode's deviation -- profiles/r02/bf16_ablation_emulated.txt) is run with the rounding enabled for ONE stage at a time
(Cfg(act_stages=[...])), and with it enabled everywhere EXCEPT one stage, against the plain f32 oracle.

    python tools/bf16_stage_ablation.py [--batch 2] [--sections 2] > profiles/r03/bf16_stage_ablation.txt
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from oracle import ast_oracle as O
from oracle import layout as OL
from oracle import seeded_params as sp

ENC_STAGES = ["in"] + [f"b{i}" for i in range(6)] + ["pool"] + [f"w.b{i}" for i in range(6)]


def run(B, S, stages, term):
    """gradient of `term` (margin | hsic | full) and the embeddings with rounding on `stages` (None = f32, 'all' = everywhere)."""
    sds = {t: OL.seeded_model_state(t) for t in ("style", "content", "decoder", "disc")}
    if stages is None:
        cfg = O.Cfg(training=True, p_drop=0.0)
    elif stages == "all":
        cfg = O.Cfg(training=True, p_drop=0.0, act_dtype=torch.bfloat16)
    else:
        cfg = O.Cfg(training=True, p_drop=0.0, act_dtype=torch.bfloat16, act_stages=stages)
    x, labels = sp.seeded_input(B, S), sp.balanced_labels(B)
    style, cls = O.style_encoder_forward(sds["style"], x, labels, cfg)
    content = O.content_encoder_forward(sds["content"], x, cfg)
    if term == "margin":
        loss = O.margin_loss(cls)
    elif term == "hsic":
        loss = O.disentanglement_loss(style, content.mean(1))
    else:
        raise KeyError(term)
    loss.backward()
    grads = {t: {k: v.grad for k, v in sds[t].items() if v.requires_grad and v.grad is not None} for t in ("style", "content")}
    return dict(style=style.detach(), cls=cls.detach(), content=content.detach(), loss=float(loss.detach()), grads=grads)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def grad_rel(ga, gb):
    num = den = 0.0
    for k, ref in gb.items():
        a = ga.get(k)
        a = torch.zeros_like(ref) if a is None else a
        num += float((a.double() - ref.double()).pow(2).sum())
        den += float(ref.double().pow(2).sum())
    return math.sqrt(num / den) if den > 0 else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--sections", type=int, default=2)
    ap.add_argument("--terms", default="margin,hsic")
    ap.add_argument("--modes", default="only,except")
    args = ap.parse_args()
    B, S = args.batch, args.sections
    torch.set_num_threads(8)
    print(f"# bf16 STORAGE rounding by stage, CPU oracle vs CPU oracle f32; B={B} S={S}; rel-L2 errors")
    print("# 'only X': rounding at stage X of BOTH encoders only; 'except X': rounding everywhere but X")
    for term in args.terms.split(","):
        t0 = time.time()
        ref = run(B, S, None, term)
        print(f"# term {term}: f32 loss {ref['loss']:.6f}   ({time.time() - t0:.1f} s per run)")
        print(f"  {'case':22s} {'loss':>10s} {'style_emb':>10s} {'c0-c1':>10s} {'content':>10s} {'g style':>10s} {'g content':>10s}", flush=True)
        cases = [("all", "all")]
        if "only" in args.modes:
            cases += [(f"only {s}", [s]) for s in ENC_STAGES]
            cases += [("only b0-b2", ["b0", "b1", "b2"]), ("only b3-b5+pool", ["b3", "b4", "b5", "pool"]), ("only weights", ["w."]),
                      ("only activations", ["in", "b", "pool"])]
        if "except" in args.modes:
            cases += [("except b4,b5,pool", ["in", "b0", "b1", "b2", "b3", "w."]), ("except b3-b5,pool", ["in", "b0", "b1", "b2", "w."]),
                      ("except b3-5,pool,w3-5", ["in", "b0", "b1", "b2", "w.b0", "w.b1", "w.b2"]),
                      ("except b2-5,pool,w2-5", ["in", "b0", "b1", "w.b0", "w.b1"]),
                      ("except pool", ["in", "b", "w."])]
        for name, st in cases:
            r = run(B, S, st, term)
            dref = ref["cls"][0] - ref["cls"][1]
            d = r["cls"][0] - r["cls"][1]
            print(f"  {name:22s} {r['loss']:10.6f} {rel(r['style'], ref['style']):10.2e} {rel(d, dref):10.2e} {rel(r['content'], ref['content']):10.2e} "
                  f"{grad_rel(r['grads']['style'], ref['grads']['style']):10.2e} {grad_rel(r['grads']['content'], ref['grads']['content']):10.2e}", flush=True)


if __name__ == "__main__":
    main()

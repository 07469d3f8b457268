"""CPU oracle of ONE optimiser step of the reconstructed train2 step: encoders -> discriminator phase
(adversarial_loss on detached embeddings, clip, Adam on D) -> generator phase (teacher-forced decoder + recon /
InfoNCE / margin / HSIC / adversarial-generator terms, clip, Adam on encoders + decoder).

TEST INFRASTRUCTURE ONLY -- NOT PRODUCT CODE.  Imported by tests/ and by bench.py's `cpu_baseline` leg; the product
package never imports anything under oracle/.

The arithmetic of every module is oracle/ast_oracle.py (pinned against the reference by tests/test_oracle_golden.py);
this file only sequences it the way ast_amd/train.py does, with torch.optim.Adam and clip_grad_norm_ as the optimiser.
`train2.ipynb` is not in the reference checkout, so the step itself has no reference fixture: the order
(D step before the generator's adversarial term), unit loss weights and max_norm 1.0 are the documented reconstruction.
"""
from __future__ import annotations

import torch

from . import ast_oracle as O
from . import layout as OL


class OracleTrainer:
    """State dicts of the four models + two Adam optimisers; step(x, labels) returns the loss scalars of the step."""

    TERMS = ("rec", "nce", "hsic", "margin", "adv_g")

    def __init__(self, lr_g=1e-4, lr_d=1e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, p_drop=0.0, act_dtype=None):
        self.sds = {t: OL.seeded_model_state(t) for t in ("style", "content", "decoder", "disc")}
        self.gparams = [v for t in ("style", "content", "decoder") for v in self.sds[t].values() if v.requires_grad]
        self.dparams = [v for v in self.sds["disc"].values() if v.requires_grad]
        self.og = torch.optim.Adam(self.gparams, lr=lr_g, betas=betas, eps=eps)
        self.od = torch.optim.Adam(self.dparams, lr=lr_d, betas=betas, eps=eps)
        self.max_grad_norm = max_grad_norm
        self.cfg = O.Cfg(training=True, p_drop=p_drop, act_dtype=act_dtype)
        self.gnorm = None

    def forward_losses(self, x, labels, with_d_step=True, rec_kwargs=None):
        """Encoders, D phase (optionally with D's optimiser step) and every generator-phase loss term, no G backward.
        Returns (dict of terms incl. 'adv_d', decoder output)."""
        sds, cfg = self.sds, self.cfg
        y = x[..., :513]
        style, cls = O.style_encoder_forward(sds["style"], x, labels, cfg)
        content = O.content_encoder_forward(sds["content"], x, cfg)
        self.od.zero_grad()
        d_loss, _ = O.adversarial_loss(sds["disc"], style.detach(), cls.detach(), content.detach(), labels, True)
        if with_d_step:
            d_loss.backward()
            if self.max_grad_norm > 0:
                torch.nn.utils.clip_grad_norm_(self.dparams, self.max_grad_norm)
            self.od.step()
        out = O.decoder_forward(sds["decoder"], content, cls[labels], cfg, y=y)
        rec = O.comprehensive_loss(out, y, **(rec_kwargs or {}))
        terms = {"rec": rec["total_loss"], "nce": O.infonce_loss(style, labels), "margin": O.margin_loss(cls),
                 "hsic": O.disentanglement_loss(style, content.mean(1)),
                 "adv_g": O.adversarial_loss(sds["disc"], style, cls, content, labels, False)[1],
                 "adv_d": d_loss.detach()}
        self.rec_parts = {k: v.detach() for k, v in rec.items()}
        self.embeddings = (style.detach(), cls.detach(), content.detach())
        return terms, out

    def step(self, x, labels, apply_g=True, use=TERMS):
        """`use`: the generator-phase terms that enter the total (TrainConfig's use_hsic / use_nce / use_adv gates)."""
        terms, _ = self.forward_losses(x, labels)
        self.og.zero_grad()
        total = None
        for k in ("rec", "nce", "margin", "hsic", "adv_g"):          # (fixed order: the total's last bits are part of the fixtures)
            if k in use:
                total = terms[k] if total is None else total + terms[k]
        total.backward()
        # the raw (unclipped) generator gradient by model and parameter name, for gradient parity checks
        self.raw_grads = {t: {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.sds[t].items() if v.requires_grad}
                          for t in ("style", "content", "decoder")}
        if self.max_grad_norm > 0:
            self.gnorm = float(torch.nn.utils.clip_grad_norm_(self.gparams, self.max_grad_norm))
        if apply_g:
            self.og.step()
        res = {k: float(v.detach()) for k, v in terms.items()}
        res["total"] = float(total.detach())
        return res


"""Drop-in for the reference's losses.py on libast_hip (single-workgroup
wavefront-reduction kernels; value and gradient from one launch each)."""
from __future__ import annotations

import torch

from . import ops


def _labels32(labels, device):
    if labels.is_cuda:
        return labels.to(torch.int32)
    return ops.const_tensor(tuple(int(v) for v in labels.tolist()), torch.int32, device)


def infoNCE_loss(style_emb: torch.Tensor, labels: torch.Tensor, temperature: float = 0.1) -> torch.Tensor:
    """losses.py:9-36."""
    return ops.InfoNCEFn.apply(style_emb, _labels32(labels, style_emb.device), float(temperature))


def margin_loss(class_emb: torch.Tensor, margin: float = 2.0, weight: float = 1.0) -> torch.Tensor:
    """losses.py:45-57 (`weight` is unused there too)."""
    return ops.MarginFn.apply(class_emb, float(margin))


def adversarial_loss(style_emb, class_emb, content_emb, discriminator, labels, compute_for_discriminator: bool,
                     lambda_content: float = 1.0, lambda_class: float = 0.5, lambda_style: float = 1.0):
    """losses.py:69-123."""
    dev = style_emb.device
    if content_emb.dim() == 3:
        content_emb = ops.mean_over_sections(content_emb)
    lab = _labels32(labels, dev)
    style_pred = discriminator(style_emb)
    content_pred = discriminator(content_emb)
    def wt(w, t):                         # a weight of exactly 1.0 needs no multiply launch
        return t if w == 1.0 else w * t
    d_loss = wt(lambda_style, ops.CrossEntropyFn.apply(style_pred, lab)) + wt(lambda_content, ops.CrossEntropyFn.apply(content_pred, lab))
    if class_emb is not None:
        class_pred = discriminator(class_emb)
        d_loss = d_loss + lambda_class * ops.CrossEntropyFn.apply(class_pred, ops.const_tensor((0, 1), torch.int32, dev))
    if compute_for_discriminator:
        return d_loss, None
    return d_loss, -wt(lambda_content, ops.SoftmaxEntropyFn.apply(content_pred))


def disentanglement_loss(style_emb: torch.Tensor, content_emb: torch.Tensor, use_hsic: bool = True, weight=20.0) -> torch.Tensor:
    """losses.py:138-191 (HSIC with the median heuristic, or the cross-covariance penalty; `weight` is unused there too)."""
    if not use_hsic:
        return ops.CrossCovFn.apply(style_emb, content_emb)
    return ops.HSICFn.apply(style_emb, content_emb)

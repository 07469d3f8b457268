"""Drop-in for the reference's content_encoder.py (ContentEncoder) on libast_hip."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import layers as L
from . import config, ops, tokprog
from .style_encoder import (CHANNELS, ResBlock, SinusoidalPositionalEncoding, _module_bank, build_resnet,
                            initialize_weights, run_resnet)  # noqa: F401


class ContentEncoder(nn.Module):
    """content_encoder.py:9-99: same CNN (`cnn.<i>.*` keys, no `.net`), `proj`, no CLS token, returns (B,S,d)."""

    def __init__(self, in_channels: int = 2, cnn_out_dim: int = 256, transformer_dim: int = 256, num_heads: int = 4,
                 num_layers: int = 4, channels_list: list = list(CHANNELS)):
        super().__init__()
        self.cnn, last = build_resnet(in_channels, list(channels_list))
        self.proj = nn.Linear(last, cnn_out_dim)
        self.input_proj = nn.Linear(cnn_out_dim, transformer_dim) if cnn_out_dim != transformer_dim else None
        self.pos_encoder = SinusoidalPositionalEncoding(transformer_dim)
        self.norm = nn.LayerNorm(transformer_dim)
        layer = nn.TransformerEncoderLayer(d_model=transformer_dim, nhead=num_heads, dim_feedforward=transformer_dim * 4,
                                           dropout=0.1, batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers)

    def register(self, bank):
        for m in self.cnn:
            if isinstance(m, ResBlock):
                m.register(bank)
        self._proj = bank.add(self.proj.weight, "linear", L.tok_dtype, bias=self.proj.bias)
        self._inp = bank.add(self.input_proj.weight, "linear", L.tok_dtype, bias=self.input_proj.bias) if self.input_proj else None
        self._layers = [L.EncoderLayer(bank, l) for l in self.transformer.layers]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, S, C, T, F = x.shape
        bank = _module_bank(self)
        bank.prepare(self.training)
        feat = run_resnet(self.cnn, ops.cached_nhwc(x, L.img_dtype()), self.training)
        feat = L.linear(feat, self._proj)
        if self._inp is not None:
            feat = L.linear(feat, self._inp)
        seq = L.layer_norm(self.pos_encoder(feat.view(B, S, -1)), self.norm)
        if config.tok_programs > 0 and tokprog.encoder_stack_ok(seq, self._layers):
            seq = tokprog.encoder_stack(seq, self._layers, self.training, xcd=1)
        else:
            for lyr in self._layers:
                seq = lyr(seq, self.training)
        return seq

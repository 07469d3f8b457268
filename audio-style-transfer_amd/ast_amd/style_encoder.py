"""Drop-in for the reference's style_encoder.py (StyleEncoder / DeepCNN / ResBlock /
SinusoidalPositionalEncoding / initialize_weights) on libast_hip.

nn.* layers are parameter containers only (identical construction order, default
init and state_dict keys: `cnn.net.0.conv1.weight_orig`, `..weight_u`, ... ); the
forward pass is NHWC implicit-GEMM convolutions + fused norm passes (ops.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import layers as L
from . import config, ops, tokprog

CHANNELS = (32, 64, 128, 256, 512, 512)


class SinusoidalPositionalEncoding(nn.Module):
    """style_encoder.py:9-29 -- buffer `pe` (1, max_len, d); forward adds pe[:, :L]."""

    def __init__(self, hidden_dim: int, max_len: int = 500):
        super().__init__()
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, hidden_dim, 2, dtype=torch.float32) * (-math.log(10000.0) / hidden_dim))
        table = torch.zeros(max_len, hidden_dim)
        table[:, 0::2], table[:, 1::2] = torch.sin(pos * freq), torch.cos(pos * freq)
        self.register_buffer("pe", table[None])

    def forward(self, x):
        return x + self.pe[:, :x.size(1)]


class ResBlock(nn.Module):
    """style_encoder.py:41-84: SN-conv3x3(s)/BN/ReLU/SN-conv3x3/BN + SN-conv1x1(s)/InstanceNorm shortcut."""

    def __init__(self, in_channels: int, out_channels: int, downsample: bool = False):
        super().__init__()
        self.stride = 2 if downsample else 1
        mk = lambda ci, co, k, s, p: spectral_norm(nn.Conv2d(ci, co, k, stride=s, padding=p))  # noqa: E731
        self.conv1 = mk(in_channels, out_channels, 3, self.stride, 1)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = mk(out_channels, out_channels, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        if downsample or in_channels != out_channels:
            self.downsample = nn.Sequential(mk(in_channels, out_channels, 1, self.stride, 0),
                                            nn.InstanceNorm2d(out_channels, affine=True))
        else:
            self.downsample = nn.Identity()

    def register(self, bank: L.WeightBank):
        if isinstance(self.downsample, nn.Identity):
            raise NotImplementedError("identity shortcut: not on the reference's configured path (all 6 blocks downsample)")
        add = lambda c: bank.add(c.weight_orig, "conv", L.img_dtype, u=c.weight_u, v=c.weight_v, bias=c.bias)  # noqa: E731
        self._pw = (add(self.conv1), add(self.conv2), add(self.downsample[0]))

    def run(self, x, training):
        p1, p2, pd = self._pw
        c1, idn, c1_stats, in_stats = L.res_head(x, p1, pd, self.stride, training)
        h = L.bn_act(c1, self.bn1, training, True, c1_stats)
        c2, c2_stats = L.conv_with_stats(h, p2, 3, 1, 1, training)
        inn = self.downsample[1]
        return ops.ResTailFn.apply(c2, idn, self.bn2.weight, self.bn2.bias, inn.weight, inn.bias, self.bn2, inn, training, c2_stats, in_stats)


def build_resnet(in_channels, channels_list):
    mods, prev = [], in_channels
    for c in channels_list:
        mods.append(ResBlock(prev, c, downsample=True))   # downsample_number=100 > len(list): every block strides
        prev = c
    mods += [nn.AdaptiveAvgPool2d((2, 5)), nn.AdaptiveAvgPool2d((1, 1))]
    return nn.Sequential(*mods), prev


def run_resnet(net: nn.Sequential, x_nhwc, training):
    """blocks -> AdaptiveAvgPool2d((2,5)) -> AdaptiveAvgPool2d((1,1)) -> (N, C) f32 tokens."""
    h = x_nhwc
    for m in net:
        if isinstance(m, ResBlock):
            h = m.run(h, training)
        else:
            h = ops.AdaptivePoolFn.apply(h, *m.output_size)
    return ops.CastFn.apply(h.reshape(h.shape[0], -1), torch.float32)


class DeepCNN(nn.Module):
    """style_encoder.py:95-129."""

    def __init__(self, in_channels: int, out_dim: int, channels_list=CHANNELS):
        super().__init__()
        self.net, last = build_resnet(in_channels, list(channels_list))
        self.proj = nn.Linear(last, out_dim)

    def register(self, bank):
        for m in self.net:
            if isinstance(m, ResBlock):
                m.register(bank)
        self._proj = bank.add(self.proj.weight, "linear", L.tok_dtype, bias=self.proj.bias)

    def run(self, x_nhwc, training):
        return L.linear(run_resnet(self.net, x_nhwc, training), self._proj)

    def forward(self, x):            # (N,2,T,F) f32 NCHW, as the reference
        bank = _module_bank(self)
        bank.prepare(self.training)
        return self.run(ops.nchw_to_nhwc(x, L.img_dtype()), self.training)


def _module_bank(mod):
    bank = mod.__dict__.get("_bank")
    if bank is None:
        bank = L.WeightBank()
        mod.register(bank)
        mod.__dict__["_bank"] = bank
    return bank


def class_prototypes(style_emb, labels):
    """style_encoder.py:243-253: mean embedding per PRESENT class id, ascending.  `labels` on the
    host avoids the device sync of labels.unique() (and is required under hipGraph capture)."""
    return ops.class_means(style_emb, labels)


class StyleEncoder(nn.Module):
    """style_encoder.py:147-258."""

    def __init__(self, in_channels: int = 2, cnn_out_dim: int = 256, transformer_dim: int = 256, num_heads: int = 4,
                 num_layers: int = 4, use_cls: bool = True):
        super().__init__()
        self.use_cls = use_cls
        self.cnn = DeepCNN(in_channels, cnn_out_dim)
        self.input_proj = nn.Linear(cnn_out_dim, transformer_dim) if cnn_out_dim != transformer_dim else None
        self.pos_encoder = SinusoidalPositionalEncoding(transformer_dim)
        self.norm = nn.LayerNorm(transformer_dim)
        layer = nn.TransformerEncoderLayer(d_model=transformer_dim, nhead=num_heads, dim_feedforward=transformer_dim * 4,
                                           dropout=0.1, batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers)
        if use_cls:
            self.cls_token = nn.Parameter(torch.randn(1, 1, transformer_dim))

    def register(self, bank):
        self.cnn.register(bank)
        self._inp = bank.add(self.input_proj.weight, "linear", L.tok_dtype, bias=self.input_proj.bias) if self.input_proj else None
        self._layers = [L.EncoderLayer(bank, l) for l in self.transformer.layers]

    def forward(self, x: torch.Tensor, labels: torch.Tensor = None):
        B, S, C, T, F = x.shape
        bank = _module_bank(self)
        bank.prepare(self.training)
        feat = self.cnn.run(ops.cached_nhwc(x, L.img_dtype()), self.training)
        if self._inp is not None:
            feat = L.linear(feat, self._inp)
        seq = feat.view(B, S, -1)
        if self.use_cls:
            seq = torch.cat([self.cls_token.expand(B, -1, -1), seq], dim=1)
        seq = L.layer_norm(self.pos_encoder(seq), self.norm)
        if config.tok_programs > 0 and tokprog.encoder_stack_ok(seq, self._layers):
            seq = tokprog.encoder_stack(seq, self._layers, self.training, xcd=0)
        else:
            for lyr in self._layers:
                seq = lyr(seq, self.training)
        style_emb = seq[:, 0, :] if self.use_cls else seq.mean(dim=1)
        class_emb = class_prototypes(style_emb, labels) if labels is not None else None
        return style_emb, class_emb


def initialize_weights(model):
    """style_encoder.py:263-308: Kaiming-normal convs (on weight_orig), Xavier-normal gain 0.2 for
    Linear / attention projections, unit BN/IN gammas.  (The cls_token branch there never fires:
    named_modules() yields no nn.Parameter.)"""
    for _, m in model.named_modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(getattr(m, "weight_orig", m.weight), mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
            if getattr(m, "weight", None) is not None:
                nn.init.ones_(m.weight)
            if getattr(m, "bias", None) is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight, gain=0.2)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.TransformerEncoderLayer):
            for w in (m.self_attn.in_proj_weight, m.self_attn.out_proj.weight, m.linear1.weight, m.linear2.weight):
                nn.init.xavier_normal_(w, gain=0.2)
            for b in (m.self_attn.in_proj_bias, m.self_attn.out_proj.bias, m.linear1.bias, m.linear2.bias):
                nn.init.zeros_(b)

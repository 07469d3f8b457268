"""CPU oracle: a functional fp32 restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this file; the product package
(audio-style-transfer_amd/) never does and fails loudly without its HIP library.

Every function works on a flat ``state_dict`` (the reference's own key layout,
e.g. ``cnn.net.0.conv1.weight_orig``) instead of nn.Module objects, and spells
out the arithmetic the reference delegates to torch.nn layers (batch/instance/
layer norm, spectral norm power iteration, multi-head attention, adaptive
pooling bins, bilinear resampling).  torch is used for dense linear algebra
(conv2d / conv_transpose2d / matmul) and for autograd only.

Pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tools/make_golden.py produced by running the real
reference modules (/root/reference) in the build container on the same seeded
parameters and inputs.  CQT (librosa) is the one piece that is *parity
unpinned* -- see oracle/frontend_oracle.py.

Citations are file:line into the reference checkout.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

EPS_BN = 1e-5
EPS_LN = 1e-5
EPS_SN = 1e-12
BN_MOMENTUM = 0.1


class Cfg:
    """Run-time switches.  `training` mirrors nn.Module.training; `p_drop` is the
    dropout probability used where the reference has nn.Dropout(0.1) -- parity
    runs use 0.0 because torch's RNG stream is not part of the contract."""

    def __init__(self, training: bool = True, p_drop: float = 0.0, act_dtype=None, act_stages=None):
        self.training = training
        self.p_drop = p_drop
        # None: the fp32 reference arithmetic.  torch.bfloat16: additionally ROUND every image activation (and its
        # gradient) and every packed conv weight to bf16 at the points where the HIP bf16 compute mode stores them
        # (conv outputs, BN(+ReLU) / ResBlock-tail / pooling outputs; accumulation, statistics, token tensors and
        # weight gradients stay f32) -- an emulation of that mode's storage precision, used to tell rounding that is
        # inherent to bf16 storage from kernel error (tests/test_gpu_bench_config.py, tools/bf16_grad_ablation.py).
        self.act_dtype = act_dtype
        # None: every storage point rounds.  A collection of stage-name prefixes ("enc.b3", "enc.pool", "dec.enc", "w.enc.b0" ...):
        # only the storage points whose tag starts with one of them round (tools/bf16_stage_ablation.py localises which stage's
        # storage rounding produces the embedding / gradient error of the bf16 mode).  `self.scope` is the model-level prefix the
        # encoder / decoder entry points set ("style", "content", "dec").
        self.act_stages = None if act_stages is None else tuple(act_stages)
        self.scope = ""

    def rounds(self, stage):
        if self.act_dtype is None:
            return False
        if self.act_stages is None:
            return True
        tag = f"{self.scope}.{stage}" if stage else self.scope
        return any(tag.startswith(p) or (stage or "").startswith(p) for p in self.act_stages)


class _RoundST(torch.autograd.Function):
    """y = round_to(dtype)(x) forward; the incoming gradient is rounded the same way backward (both are stored in
    that dtype by the emulated path)."""

    @staticmethod
    def forward(ctx, x, dtype, round_grad):
        ctx.dtype, ctx.round_grad = dtype, round_grad
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.round_grad else g), None, None


def _q(x, cfg: Cfg, stage=None):
    """activation storage point"""
    return _RoundST.apply(x, cfg.act_dtype, True) if cfg.rounds(stage) else x


def _qw(w, cfg: Cfg, stage=None):
    """packed-weight storage point (the weight GRADIENT is accumulated in f32: no rounding backward)"""
    return _RoundST.apply(w, cfg.act_dtype, False) if cfg.rounds("w." + (stage or "")) else w


def _drop(x, cfg: Cfg):
    if cfg.training and cfg.p_drop > 0.0:
        return F.dropout(x, cfg.p_drop, True)
    return x


# --------------------------------------------------------------------------
# spectral norm  (torch/nn/utils/spectral_norm.py:92-114; call sites
# style_encoder.py:50,57,67 and new_decoder.py:29-96)
# --------------------------------------------------------------------------
def spectral_weight(sd, prefix: str, cfg: Cfg, dim: int = 0):
    w = sd[prefix + "weight_orig"]
    u = sd[prefix + "weight_u"]
    v = sd[prefix + "weight_v"]
    wm = w if dim == 0 else w.transpose(0, dim)
    wm = wm.reshape(wm.shape[0], -1)
    if cfg.training:
        with torch.no_grad():
            t = wm.t() @ u
            v.copy_(t / t.norm().clamp_min(EPS_SN))
            s = wm @ v
            u.copy_(s / s.norm().clamp_min(EPS_SN))
    uu, vv = u.detach().clone(), v.detach().clone()
    sigma = torch.dot(uu, wm @ vv)
    return w / sigma


def conv_sn(sd, prefix, x, cfg, stride=1, padding=0, stage=None):
    return _q(F.conv2d(x, _qw(spectral_weight(sd, prefix, cfg), cfg, stage), sd[prefix + "bias"], stride=stride, padding=padding), cfg, stage)


def convT_sn(sd, prefix, x, cfg, stride=1, padding=0, output_padding=0, stage=None):
    w = _qw(spectral_weight(sd, prefix, cfg, dim=1), cfg, stage)
    return _q(F.conv_transpose2d(x, w, sd[prefix + "bias"], stride=stride, padding=padding,
                                 output_padding=output_padding), cfg, stage)


# --------------------------------------------------------------------------
# normalisation layers
# --------------------------------------------------------------------------
def batchnorm2d(sd, prefix, x, cfg):
    """nn.BatchNorm2d: batch statistics in training (biased var to normalise,
    unbiased var into running_var, momentum 0.1); running stats in eval."""
    g, b = sd[prefix + "weight"], sd[prefix + "bias"]
    if cfg.training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = x.numel() // x.shape[1]
            sd[prefix + "running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            sd[prefix + "running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / max(n - 1, 1))
            sd[prefix + "num_batches_tracked"].add_(1)
    else:
        mean, var = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + EPS_BN)
    return xh * g[None, :, None, None] + b[None, :, None, None]


def instancenorm2d(sd, prefix, x):
    """nn.InstanceNorm2d(affine=True, track_running_stats=False) -- style_encoder.py:69."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    xh = (x - mean) * torch.rsqrt(var + EPS_BN)
    return xh * sd[prefix + "weight"][None, :, None, None] + sd[prefix + "bias"][None, :, None, None]


def layernorm(sd, prefix, x):
    mean = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    return (x - mean) * torch.rsqrt(var + EPS_LN) * sd[prefix + "weight"] + sd[prefix + "bias"]


def linear(sd, prefix, x):
    return x @ sd[prefix + "weight"].t() + sd[prefix + "bias"]


# --------------------------------------------------------------------------
# pooling / resampling with the exact index rules
# --------------------------------------------------------------------------
def _adaptive_bins(n_in, n_out):
    return [(int(math.floor(i * n_in / n_out)), int(math.ceil((i + 1) * n_in / n_out))) for i in range(n_out)]


def adaptive_avg_pool2d(x, out_hw):
    """nn.AdaptiveAvgPool2d: bin i covers [floor(i*In/Out), ceil((i+1)*In/Out))."""
    H, W = x.shape[-2:]
    Ph = torch.zeros(out_hw[0], H, dtype=x.dtype)
    for i, (s, e) in enumerate(_adaptive_bins(H, out_hw[0])):
        Ph[i, s:e] = 1.0 / (e - s)
    Pw = torch.zeros(out_hw[1], W, dtype=x.dtype)
    for j, (s, e) in enumerate(_adaptive_bins(W, out_hw[1])):
        Pw[j, s:e] = 1.0 / (e - s)
    return torch.einsum("ih,nchw,jw->ncij", Ph, x, Pw)


def _bilinear_matrix(n_in, n_out, dtype):
    """nn.Upsample(mode='bilinear', align_corners=False): src=(dst+0.5)*in/out-0.5,
    clamped at 0; taps floor(src) and min(floor(src)+1, in-1)."""
    M = torch.zeros(n_out, n_in, dtype=dtype)
    scale = n_in / n_out
    for o in range(n_out):
        src = max((o + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        lam = src - i0
        M[o, i0] += 1.0 - lam
        M[o, i1] += lam
    return M


def bilinear_resize(x, out_hw):
    Mh = _bilinear_matrix(x.shape[-2], out_hw[0], x.dtype)
    Mw = _bilinear_matrix(x.shape[-1], out_hw[1], x.dtype)
    return torch.einsum("oh,nchw,pw->ncop", Mh, x, Mw)


# --------------------------------------------------------------------------
# attention / transformer layers (torch defaults the reference relies on)
# --------------------------------------------------------------------------
def mha(sd, prefix, q_in, kv_in, nhead, cfg, causal=False):
    """nn.MultiheadAttention(batch_first=True) with packed in_proj."""
    d = q_in.shape[-1]
    dh = d // nhead
    w, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = q_in @ w[:d].t() + b[:d]
    k = kv_in @ w[d:2 * d].t() + b[d:2 * d]
    v = kv_in @ w[2 * d:].t() + b[2 * d:]
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, nhead, dh).transpose(1, 2)
    k = k.view(B, Lk, nhead, dh).transpose(1, 2)
    v = v.view(B, Lk, nhead, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        mask = torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), diagonal=1)
        s = s.masked_fill(mask, float("-inf"))
    p = _drop(torch.softmax(s, dim=-1), cfg)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, d)
    return o @ sd[prefix + "out_proj.weight"].t() + sd[prefix + "out_proj.bias"]


def encoder_layer(sd, prefix, x, nhead, cfg):
    """nn.TransformerEncoderLayer defaults: post-norm, ReLU (style_encoder.py:181-187)."""
    x = layernorm(sd, prefix + "norm1.", x + _drop(mha(sd, prefix + "self_attn.", x, x, nhead, cfg), cfg))
    ff = linear(sd, prefix + "linear2.", _drop(torch.relu(linear(sd, prefix + "linear1.", x)), cfg))
    return layernorm(sd, prefix + "norm2.", x + _drop(ff, cfg))


def decoder_layer(sd, prefix, x, memory, nhead, cfg):
    """nn.TransformerDecoderLayer(norm_first=True) (new_decoder.py:111-118)."""
    h = layernorm(sd, prefix + "norm1.", x)
    x = x + _drop(mha(sd, prefix + "self_attn.", h, h, nhead, cfg, causal=True), cfg)
    h = layernorm(sd, prefix + "norm2.", x)
    x = x + _drop(mha(sd, prefix + "multihead_attn.", h, memory, nhead, cfg), cfg)
    h = layernorm(sd, prefix + "norm3.", x)
    ff = linear(sd, prefix + "linear2.", _drop(torch.relu(linear(sd, prefix + "linear1.", h)), cfg))
    return x + _drop(ff, cfg)


def positional_encoding(L, d, dtype=torch.float32):
    """style_encoder.py:9-29."""
    pos = torch.arange(L, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe = torch.zeros(L, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


# --------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------
def resblock(sd, prefix, x, cfg, stride=2, stage=None):
    """style_encoder.py:41-84."""
    idn = conv_sn(sd, prefix + "downsample.0.", x, cfg, stride=stride, padding=0, stage=stage)
    idn = instancenorm2d(sd, prefix + "downsample.1.", idn)
    out = conv_sn(sd, prefix + "conv1.", x, cfg, stride=stride, padding=1, stage=stage)
    out = _q(torch.relu(batchnorm2d(sd, prefix + "bn1.", out, cfg)), cfg, stage)
    out = conv_sn(sd, prefix + "conv2.", out, cfg, stride=1, padding=1, stage=stage)
    out = batchnorm2d(sd, prefix + "bn2.", out, cfg)
    return _q(torch.relu(out + idn), cfg, stage)


def deep_cnn(sd, net_prefix, proj_prefix, x, cfg, nblocks=6, return_blocks=False):
    """style_encoder.py:95-129 / content_encoder.py:22-46,80-85."""
    feats = []
    x = _q(x, cfg, "in")
    for i in range(nblocks):
        x = resblock(sd, f"{net_prefix}{i}.", x, cfg, stage=f"b{i}")
        feats.append(x)
    x = _q(adaptive_avg_pool2d(x, (2, 5)), cfg, "pool")
    x = _q(adaptive_avg_pool2d(x, (1, 1)), cfg, "pool").flatten(1)
    out = linear(sd, proj_prefix, x)
    return (out, feats) if return_blocks else out


def _encoder_stack(sd, seq, nhead, nlayers, cfg):
    for i in range(nlayers):
        seq = encoder_layer(sd, f"transformer.layers.{i}.", seq, nhead, cfg)
    return seq


def style_encoder_forward(sd, x, labels: Optional[torch.Tensor], cfg, nhead=4, nlayers=4):
    """style_encoder.py:199-258."""
    B, S = x.shape[:2]
    cfg.scope = "style"
    feat = deep_cnn(sd, "cnn.net.", "cnn.proj.", x.reshape(B * S, *x.shape[2:]), cfg)
    seq = feat.view(B, S, -1)
    seq = torch.cat([sd["cls_token"].expand(B, -1, -1), seq], dim=1)
    seq = seq + positional_encoding(seq.shape[1], seq.shape[2])
    seq = layernorm(sd, "norm.", seq)
    enc = _encoder_stack(sd, seq, nhead, nlayers, cfg)
    style = enc[:, 0, :]
    if labels is None:
        return style, None
    rows = [style[labels == c].mean(0) for c in sorted(set(labels.tolist()))]
    return style, torch.stack(rows, 0)


def content_encoder_forward(sd, x, cfg, nhead=4, nlayers=4):
    """content_encoder.py:70-99."""
    B, S = x.shape[:2]
    cfg.scope = "content"
    feat = deep_cnn(sd, "cnn.", "proj.", x.reshape(B * S, *x.shape[2:]), cfg)
    seq = feat.view(B, S, -1)
    seq = seq + positional_encoding(S, seq.shape[2])
    seq = layernorm(sd, "norm.", seq)
    return _encoder_stack(sd, seq, nhead, nlayers, cfg)


# --------------------------------------------------------------------------
# decoder (new_decoder.py)
# --------------------------------------------------------------------------
def decoder_encode_input(sd, y, cfg):
    """new_decoder.py:145-168 -- y: (N,2,287,513) -> (N,256)."""
    cfg.scope = "dec"
    h = _q(y, cfg, "in")
    for idx, stride in ((0, 1), (3, 2), (6, 2), (9, 2)):
        h = conv_sn(sd, f"conv_encoder.{idx}.", h, cfg, stride=stride, padding=1, stage=f"enc{idx}")
        h = _q(torch.relu(batchnorm2d(sd, f"conv_encoder.{idx + 1}.", h, cfg)), cfg, f"enc{idx}")
    h = _q(adaptive_avg_pool2d(h, (32, 16)), cfg, "encpool")
    h = conv_sn(sd, "spatial_projection.0.", h, cfg, stride=1, padding=1, stage="sp")
    h = _q(torch.relu(batchnorm2d(sd, "spatial_projection.1.", h, cfg)), cfg, "sp")
    h = conv_sn(sd, "spatial_projection.3.", h, cfg, stride=1, padding=0, stage="sp")
    return linear(sd, "feature_to_sequence.", h.flatten(1))


def decoder_generate_output(sd, tok, cfg):
    """new_decoder.py:170-193 -- tok: (B,S,256) -> (B,S,2,287,513)."""
    B, S, _ = tok.shape
    cfg.scope = "dec"
    h = _q(linear(sd, "sequence_to_feature.", layernorm(sd, "output_norm.", tok)).view(B * S, 1, 32, 16), cfg, "gen_in")
    for idx in (0, 3, 6, 9):
        h = convT_sn(sd, f"conv_decoder.{idx}.", h, cfg, stride=2, padding=1, output_padding=1, stage=f"gen{idx}")
        h = _q(torch.relu(batchnorm2d(sd, f"conv_decoder.{idx + 1}.", h, cfg)), cfg, f"gen{idx}")
    h = convT_sn(sd, "conv_decoder.12.", h, cfg, stride=1, padding=1, stage="gen12")
    return bilinear_resize(h, (287, 513)).view(B, S, 2, 287, 513)


def decoder_prepare_memory(sd, content, class_emb, cfg):
    """new_decoder.py:208-229."""
    S = content.shape[1]
    cm = linear(sd, "content_proj.", content)
    km = linear(sd, "class_proj.", class_emb).unsqueeze(1).expand(-1, S, -1)
    return _drop(torch.cat([cm, km], dim=1), cfg)


def _decoder_stack(sd, tgt, memory, nhead, nlayers, cfg):
    for i in range(nlayers):
        tgt = decoder_layer(sd, f"transformer_decoder.layers.{i}.", tgt, memory, nhead, cfg)
    return tgt


def decoder_forward(sd, content, class_emb, cfg, y=None, target_length=None, nhead=4, nlayers=4):
    """new_decoder.py:321-345 (teacher forcing 231-269, autoregressive 272-319)."""
    memory = decoder_prepare_memory(sd, content, class_emb, cfg)
    B = memory.shape[0]
    d = memory.shape[-1]
    if cfg.training and y is not None:
        if y.dim() != 5:
            raise ValueError(f"Expected y to have shape [B, S, 2, 287, 513], got {tuple(y.shape)}")
        S = y.shape[1]
        emb = decoder_encode_input(sd, y.reshape(B * S, *y.shape[2:]), cfg).view(B, S, d)
        tgt = torch.cat([sd["start_token"].expand(B, 1, -1), emb[:, :-1]], dim=1)
        tgt = layernorm(sd, "input_norm.", tgt + positional_encoding(S, d))
        return decoder_generate_output(sd, _decoder_stack(sd, tgt, memory, nhead, nlayers, cfg), cfg)
    if target_length is None:
        target_length = memory.shape[1] // 2
    seq = sd["start_token"].expand(B, -1, -1)
    outs = []
    for _ in range(target_length):
        cur = seq + positional_encoding(seq.shape[1], d)  # no input_norm here (new_decoder.py:296)
        nxt = _decoder_stack(sd, cur, memory, nhead, nlayers, cfg)[:, -1:, :]
        outs.append(nxt)
        seq = torch.cat([seq, nxt], dim=1)
    return decoder_generate_output(sd, torch.cat(outs, dim=1), cfg)


def simple_decoder_forward(sd, content, class_emb, cfg, y=None, target_length=None, nhead=4, nlayers=4):
    """SimpleDecoder_TransformerOnly.py:127-133 (teacher forcing :80-102, autoregressive :104-125): new_decoder's
    transformer with the CNN halves replaced by two 2*287*513 x 256 linears (:16-17, :56-66)."""
    memory = decoder_prepare_memory(sd, content, class_emb, cfg)          # same projections + dropout (:72-78)
    B, d = memory.shape[0], memory.shape[-1]

    def generate(tok):
        Bq, S, _ = tok.shape
        return linear(sd, "embedding_to_stft.", layernorm(sd, "output_norm.", tok)).reshape(Bq, S, 2, 287, 513)

    if cfg.training and y is not None:
        S = y.shape[1]
        emb = linear(sd, "stft_to_embedding.", y.reshape(B * S, -1)).view(B, S, d)
        tgt = torch.cat([sd["start_token"].expand(B, 1, -1), emb[:, :-1]], dim=1)
        tgt = layernorm(sd, "input_norm.", tgt + positional_encoding(S, d))
        return generate(_decoder_stack(sd, tgt, memory, nhead, nlayers, cfg))
    if target_length is None:
        target_length = memory.shape[1] // 2
    seq = sd["start_token"].expand(B, -1, -1)
    outs = []
    for _ in range(target_length):
        cur = seq + positional_encoding(seq.shape[1], d)
        nxt = _decoder_stack(sd, cur, memory, nhead, nlayers, cfg)[:, -1:, :]
        outs.append(nxt)
        seq = torch.cat([seq, nxt], dim=1)
    return generate(torch.cat(outs, dim=1))


def comprehensive_loss(out, tgt, lambda_temporal=0.3, lambda_phase=0.2, lambda_spectral=0.1, mse_weight=2.0):
    """new_decoder.py:348-420.  dim 1 = sections ("temporal"), dim 3 = the 287
    axis ("spectral").  SimpleDecoder_TransformerOnly.py:136-204 is the same function with the MSE term weighted
    1.0 instead of 2.0 (:194 vs new_decoder.py:406)."""
    mse = ((out - tgt) ** 2).mean()
    mo = torch.sqrt(out[:, :, 0] ** 2 + out[:, :, 1] ** 2 + 1e-8)
    mt = torch.sqrt(tgt[:, :, 0] ** 2 + tgt[:, :, 1] ** 2 + 1e-8)
    mag = ((mo - mt) ** 2).mean()
    dphi = torch.atan2(out[:, :, 1], out[:, :, 0]) - torch.atan2(tgt[:, :, 1], tgt[:, :, 0])
    dphi = torch.remainder(dphi + math.pi, 2 * math.pi) - math.pi
    phase = (dphi ** 2).mean()
    if out.shape[1] > 1:
        temporal = (((out[:, 1:] - out[:, :-1]) - (tgt[:, 1:] - tgt[:, :-1])) ** 2).mean()
    else:
        temporal = torch.zeros(())
    spectral = (((out[:, :, :, 1:] - out[:, :, :, :-1]) - (tgt[:, :, :, 1:] - tgt[:, :, :, :-1])) ** 2).mean()
    total = mse_weight * mse + 0.5 * mag + lambda_phase * phase + lambda_temporal * temporal + lambda_spectral * spectral
    return {"total_loss": total, "mse_loss": mse, "mag_loss": mag, "phase_loss": phase,
            "temporal_loss": temporal, "spectral_loss": spectral}


# --------------------------------------------------------------------------
# discriminator + losses (discriminator.py, losses.py)
# --------------------------------------------------------------------------
def discriminator_forward(sd, e):
    """discriminator.py:14-28."""
    h = torch.relu(linear(sd, "net.0.", e))
    h = torch.relu(linear(sd, "net.2.", h))
    return linear(sd, "net.4.", h)


def _cross_entropy(logits, target):
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.gather(1, target[:, None]).squeeze(1)).mean()


def infonce_loss(style, labels, temperature=0.1):
    """losses.py:9-36."""
    e = style / style.norm(dim=1, keepdim=True).clamp_min(1e-12)
    sim = e @ e.t()
    B = sim.shape[0]
    eye = torch.eye(B, dtype=torch.bool)
    logits = sim.masked_fill(eye, -1e9) / temperature
    logp = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    pos = (labels[:, None] == labels[None, :]) & ~eye
    per_anchor = (logp * pos).sum(1) / pos.sum(1).clamp(min=1)
    return -per_anchor.mean()


def margin_loss(class_emb, margin=2.0):
    """losses.py:45-57."""
    C = class_emb.shape[0]
    terms = []
    for i in range(C):
        for j in range(i + 1, C):
            dist = (class_emb[i] - class_emb[j]).norm()
            terms.append(torch.relu(margin - dist) ** 2)
    return torch.stack(terms).mean()


def adversarial_loss(sd_disc, style, class_emb, content, labels, compute_for_discriminator,
                     lambda_content=1.0, lambda_class=0.5, lambda_style=1.0):
    """losses.py:69-123."""
    if content.dim() == 3:
        content = content.mean(dim=1)
    ps = discriminator_forward(sd_disc, style)
    pc = discriminator_forward(sd_disc, content)
    d_loss = lambda_style * _cross_entropy(ps, labels) + lambda_content * _cross_entropy(pc, labels)
    if class_emb is not None:
        pk = discriminator_forward(sd_disc, class_emb)
        d_loss = d_loss + lambda_class * _cross_entropy(pk, torch.tensor([0, 1]))
    if compute_for_discriminator:
        return d_loss, None
    prob = torch.softmax(pc, dim=-1)
    ent = -(prob * torch.log(prob + 1e-8)).sum(-1).mean()
    return d_loss, -lambda_content * ent


def hsic_sigma_rank(B):
    """Rank (0-based, ascending) inside the flattened 2Bx2B distance matrix that
    losses.py:170-171 ends up selecting: `dist[triu_indices(...)]` is a ROW gather
    that repeats every row 2B-1 times, so torch.median (lower median) of it is the
    lower median of the whole matrix, diagonal zeros included."""
    m = 2 * B
    n = m * m * (m - 1)
    return ((n - 1) // 2) // (m - 1)


def disentanglement_loss(style, content, use_hsic=True):
    """losses.py:138-191."""
    B = style.shape[0]
    S = style - style.mean(0, keepdim=True)
    C = content - content.mean(0, keepdim=True)
    if not use_hsic:
        return (((S.t() @ C) / (B - 1)) ** 2).sum()
    X = torch.cat([style, content], dim=0)
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    nz = d2 > 0  # cdist's backward is 0 at zero distance; keep sqrt'(0) out of the graph
    dist = torch.where(nz, torch.sqrt(torch.where(nz, d2, torch.ones_like(d2))), torch.zeros_like(d2))
    flat = dist.flatten()
    order = torch.argsort(flat.detach(), stable=True)
    sigma = flat[order[hsic_sigma_rank(B)]]
    H = torch.eye(B) - torch.full((B, B), 1.0 / B)

    def rbf(Z):
        n = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(-1)
        return torch.exp(-n / (2 * sigma ** 2))

    return torch.trace((rbf(S) @ H) @ (rbf(C) @ H)) / ((B - 1) ** 2)

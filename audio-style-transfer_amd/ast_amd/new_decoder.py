"""Drop-in for the reference's new_decoder.py (Decoder, compute_comprehensive_loss) on libast_hip."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import layers as L
from . import config, ops, tokprog
from .style_encoder import SinusoidalPositionalEncoding, _module_bank

ENC_CH = ((2, 16, 1), (16, 32, 2), (32, 64, 2), (64, 64, 2))        # (cin, cout, stride)  new_decoder.py:29-48
DEC_CH = ((1, 64), (64, 32), (32, 16), (16, 8))                      # stride-2 transposed convs  new_decoder.py:72-93


def _conv_bn_relu_slots(conv, ch):
    return [conv, nn.BatchNorm2d(ch), nn.Identity()]      # slot 3 is the (parameter-free) ReLU


class Decoder(nn.Module):
    """new_decoder.py:9-345.  Sequential slot numbering is kept so state_dict keys match
    (conv_encoder.0/1/3/4/..., spatial_projection.0/1/3, conv_decoder.0/1/.../12)."""

    def __init__(self, d_model=256, nhead=4, num_layers=4, dim_feedforward=1024, dropout=0.1, max_seq_len=1000):
        super().__init__()
        self.d_model, self.max_seq_len = d_model, max_seq_len
        self.F_compressed, self.T_compressed, self.feature_dim = 32, 16, 64
        enc = []
        for cin, cout, s in ENC_CH:
            enc += _conv_bn_relu_slots(spectral_norm(nn.Conv2d(cin, cout, kernel_size=3, stride=s, padding=1)), cout)
        enc.append(nn.AdaptiveAvgPool2d((self.F_compressed, self.T_compressed)))
        self.conv_encoder = nn.Sequential(*enc)
        fd = self.feature_dim
        self.spatial_projection = nn.Sequential(
            *_conv_bn_relu_slots(spectral_norm(nn.Conv2d(fd, fd, kernel_size=3, padding=1)), fd),
            spectral_norm(nn.Conv2d(fd, 1, kernel_size=1)))
        self.feature_to_sequence = nn.Linear(self.F_compressed * self.T_compressed, d_model)
        self.sequence_to_feature = nn.Linear(d_model, self.F_compressed * self.T_compressed)
        dec = []
        for cin, cout in DEC_CH:
            dec += _conv_bn_relu_slots(
                spectral_norm(nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1)), cout)
        dec.append(spectral_norm(nn.ConvTranspose2d(8, 2, kernel_size=3, padding=1)))
        dec.append(nn.Upsample(size=(287, 513), mode="bilinear", align_corners=False))
        self.conv_decoder = nn.Sequential(*dec)
        self.content_proj = nn.Linear(d_model, d_model)
        self.class_proj = nn.Linear(d_model, d_model)
        self.pos_encoding = SinusoidalPositionalEncoding(d_model)
        layer = nn.TransformerDecoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                                           batch_first=True, norm_first=True)
        self.transformer_decoder = nn.TransformerDecoder(layer, num_layers=num_layers)
        self.start_token = nn.Parameter(torch.randn(1, 1, d_model))
        self.input_norm = nn.LayerNorm(d_model)
        self.output_norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self._init_weights()

    def _init_weights(self):
        """new_decoder.py:134-143: Xavier-uniform(gain 0.2) for >=2-D `*weight*`, zeros for 1-D weights and biases
        (so a fresh decoder outputs exactly 0 -- SURVEY F7)."""
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p, gain=0.2) if p.dim() > 1 else nn.init.zeros_(p)
            elif "bias" in name:
                nn.init.zeros_(p)

    # ---- weight registration ------------------------------------------------------
    def register(self, bank):
        conv = lambda c: bank.add(c.weight_orig, "conv", L.img_dtype, u=c.weight_u, v=c.weight_v, bias=c.bias)  # noqa: E731
        convt = lambda c: bank.add(c.weight_orig, "convT", L.img_dtype, u=c.weight_u, v=c.weight_v, bias=c.bias)  # noqa: E731
        lin = lambda l: bank.add(l.weight, "linear", L.tok_dtype, bias=l.bias)  # noqa: E731
        self._enc = [conv(self.conv_encoder[i]) for i in (0, 3, 6, 9)]
        self._sp = [conv(self.spatial_projection[0]), conv(self.spatial_projection[3])]
        self._dec = [convt(self.conv_decoder[i]) for i in (0, 3, 6, 9, 12)]
        self._f2s, self._s2f = lin(self.feature_to_sequence), lin(self.sequence_to_feature)
        self._cp, self._kp = lin(self.content_proj), lin(self.class_proj)
        self._layers = [L.DecoderLayer(bank, l) for l in self.transformer_decoder.layers]

    def _prepare(self):
        _module_bank(self).prepare(self.training)

    # ---- CNN halves -----------------------------------------------------------------
    def _encode_nhwc(self, h):
        tr = self.training
        for pw, i, (_, _, s) in zip(self._enc, (0, 3, 6, 9), ENC_CH):
            h = L.conv_bn_act(h, pw, 3, s, 1, self.conv_encoder[i + 1], tr, relu=True)
        h = ops.AdaptivePoolFn.apply(h, self.F_compressed, self.T_compressed)
        h = L.conv_bn_act(h, self._sp[0], 3, 1, 1, self.spatial_projection[1], tr, relu=True)
        h = L.conv(h, self._sp[1], 1, 1, 0, True)                          # (N,32,16,8): channel 0 is the real one
        flat = ops.CastFn.apply(h[..., 0].reshape(h.shape[0], -1), torch.float32)
        return L.linear(flat, self._f2s)

    def encode_input(self, x):
        """new_decoder.py:145-168: (N,2,287,513) f32 -> (N,d_model)."""
        self._prepare()
        return self._encode_nhwc(ops.nchw_to_nhwc(x, L.img_dtype()))

    def _generate(self, tok):
        B, S, D = tok.shape
        tr = self.training
        h = L.linear(L.layer_norm(tok, self.output_norm).reshape(B * S, D), self._s2f)       # (N,512) f32
        h = ops.CastFn.apply(h, L.img_dtype()).view(B * S, self.F_compressed, self.T_compressed, 1)
        h = torch.cat([h, h.new_zeros(B * S, self.F_compressed, self.T_compressed, 7)], dim=3)   # pad C 1 -> 8
        for pw, i in zip(self._dec[:4], (0, 3, 6, 9)):
            h = L.convT_bn_act(h, pw, 3, 2, 1, 1, self.conv_decoder[i + 1], tr, relu=True)
        h = L.convT(h, self._dec[4], 3, 1, 1, 0, True)                                      # (N,512,256,8)
        out = ops.BilinearToNCHWFn.apply(h, 2, 287, 513)
        return out.view(B, S, 2, 287, 513)

    def generate_output(self, decoder_outputs):
        """new_decoder.py:170-193."""
        self._prepare()
        return self._generate(decoder_outputs)

    def create_causal_mask(self, seq_len):
        return torch.triu(torch.ones(seq_len, seq_len), diagonal=1).bool()

    def _memory(self, content_emb, class_emb):
        B, Sc, D = content_emb.shape
        cm = L.linear(content_emb.reshape(B * Sc, D), self._cp).view(B, Sc, D)
        km = L.linear(class_emb, self._kp).unsqueeze(1).expand(-1, Sc, -1)
        return ops.dropout(torch.cat([cm, km], dim=1), self.dropout.p, self.training)

    def prepare_memory(self, content_emb, class_emb):
        """new_decoder.py:208-229."""
        self._prepare()
        return self._memory(content_emb, class_emb)

    def _stack(self, tgt, memory):
        if config.tok_programs > 0 and tokprog.decoder_stack_ok(tgt, memory, self._layers):
            return tokprog.decoder_stack(tgt, memory, self._layers, self.training)
        for lyr in self._layers:
            tgt = lyr(tgt, memory, self.training)
        return tgt

    def encode_target(self, y):
        """Teacher-forcing embeddings of the target sections (new_decoder.py:246-248).  Independent of the
        encoders, so the trainer runs it on its own stream; pass the result to forward(y_embeddings=...)."""
        self._prepare()
        B, S = y.shape[:2]
        return self._encode_nhwc(ops.nchw_to_nhwc(y.view(B * S, *y.shape[2:]), L.img_dtype())).view(B, S, self.d_model)

    def _training_pass(self, y, memory, y_embeddings=None):
        B, S = y.shape[:2]
        emb = y_embeddings
        if emb is None:
            emb = self._encode_nhwc(ops.nchw_to_nhwc(y.view(B * S, *y.shape[2:]), L.img_dtype())).view(B, S, self.d_model)
        tgt = torch.cat([self.start_token.expand(B, 1, -1), emb[:, :-1, :]], dim=1)
        tgt = L.layer_norm(self.pos_encoding(tgt), self.input_norm)
        return self._generate(self._stack(tgt, memory))

    def forward_training(self, y, memory):
        """new_decoder.py:231-269."""
        self._prepare()
        return self._training_pass(y, memory)

    # How forward_inference walks the sequence: "recompute" = the reference's loop (new_decoder.py:294-314: the whole
    # stack over all tokens generated so far, every step); "kv_cache" = one new token per step against cached per-layer
    # K/V (identical results: the layers are causal).  Class attribute so callers and tests can switch it.
    decode_mode = "recompute"

    def _inference_pass_cached(self, memory, target_length):
        B, d = memory.size(0), self.d_model
        mem_kv = [lyr.memory_kv(memory) for lyr in self._layers]
        caches = [None] * len(self._layers)
        tok = self.start_token.expand(B, -1, -1)
        pe = self.pos_encoding.pe
        outs = []
        for t in range(target_length):
            x = tok + pe[:, t:t + 1]
            for i, lyr in enumerate(self._layers):
                x, caches[i] = lyr.step(x, caches[i], mem_kv[i])
            outs.append(x)
            tok = x
        return self._generate(torch.cat(outs, dim=1))

    def _inference_pass(self, memory, target_length=None):
        B = memory.size(0)
        if target_length is None:
            target_length = memory.size(1) // 2
        if self.decode_mode == "kv_cache" and not torch.is_grad_enabled():
            return self._inference_pass_cached(memory, target_length)
        seq = self.start_token.expand(B, -1, -1)
        outs = []
        for _ in range(target_length):
            nxt = self._stack(self.pos_encoding(seq), memory)[:, -1:, :]      # no input_norm here (new_decoder.py:296)
            outs.append(nxt)
            seq = torch.cat([seq, nxt], dim=1)
        return self._generate(torch.cat(outs, dim=1))

    def forward_inference(self, memory, target_length=None):
        """new_decoder.py:272-319."""
        self._prepare()
        return self._inference_pass(memory, target_length)

    def forward(self, content_emb, class_emb, y=None, target_length=None, y_embeddings=None):
        """new_decoder.py:321-345 (+ optional precomputed encode_target(y))."""
        if y_embeddings is None or not (self.training and y is not None):
            self._prepare()
        memory = self._memory(content_emb, class_emb)
        if self.training and y is not None:
            if len(y.shape) != 5:
                raise ValueError(f"Expected y to have shape [B, S, 2, 287, 513], got {y.shape}")
            return self._training_pass(y, memory, y_embeddings)
        return self._inference_pass(memory, target_length)


def compute_comprehensive_loss(output, target, lambda_temporal=0.3, lambda_phase=0.2, lambda_spectral=0.1):
    """new_decoder.py:348-420 as ONE streaming kernel pass (value + gradient).  Only `total_loss`
    carries gradient; the component entries are detached values."""
    return _comprehensive_loss(output, target, lambda_temporal, lambda_phase, lambda_spectral, 2.0)


def _comprehensive_loss(output, target, lambda_temporal, lambda_phase, lambda_spectral, mse_weight):
    B, S, _, Freq, T = output.shape
    n = B * S * Freq * T
    c = (mse_weight / (2 * n), 0.5 / n, lambda_phase / n,
         (lambda_temporal / (2 * B * (S - 1) * Freq * T)) if S > 1 else 0.0,
         (lambda_spectral / (2 * B * S * (Freq - 1) * T)) if Freq > 1 else 0.0)
    # the five reported components are sums * 1/count, formed by the same finishing launch as the total (they carry no gradient)
    inv = (1.0 / (2 * n), 1.0 / n, 1.0 / n,
           1.0 / (2 * B * (S - 1) * Freq * T) if S > 1 else 0.0,
           1.0 / (2 * B * S * (Freq - 1) * T) if Freq > 1 else 0.0)
    total, _, parts = ops.ReconTotalFn.apply(output.contiguous(), target, c, inv)
    return {"total_loss": total, "mse_loss": parts[0], "mag_loss": parts[1], "phase_loss": parts[2],
            "temporal_loss": parts[3], "spectral_loss": parts[4]}

"""Drop-in for the reference's SimpleDecoder_TransformerOnly.py (the decoder the shipped checkpoints / evaluation
scripts use): the CNN halves of new_decoder.Decoder replaced by two 2*287*513 x 256 linears around the
same pre-norm transformer decoder.  The two big linears stream their 301 MB f32 weights once per use through
ast_bigk_gemm / ast_skinny_gemm / ast_bign_dgrad / ast_linear_wgrad (no packed copies)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import layers as L
from . import ops
from .new_decoder import _comprehensive_loss
from .style_encoder import SinusoidalPositionalEncoding, _module_bank


class Decoder(nn.Module):
    """SimpleDecoder_TransformerOnly.py:9-133; identical attribute names and state_dict keys."""

    def __init__(self, d_model=256, nhead=4, num_layers=4, dim_feedforward=1024, dropout=0.1):
        super().__init__()
        self.d_model = d_model
        self.stft_dim = 2 * 287 * 513
        self.stft_to_embedding = nn.Linear(self.stft_dim, d_model)
        self.embedding_to_stft = nn.Linear(d_model, self.stft_dim)
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
        """SimpleDecoder_TransformerOnly.py:46-54 (same rule as new_decoder.py:134-143)."""
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p, gain=0.2) if p.dim() > 1 else nn.init.zeros_(p)
            elif "bias" in name:
                nn.init.zeros_(p)

    def register(self, bank):
        lin = lambda l: bank.add(l.weight, "linear", L.tok_dtype, bias=l.bias)  # noqa: E731
        self._cp, self._kp = lin(self.content_proj), lin(self.class_proj)
        self._layers = [L.DecoderLayer(bank, l) for l in self.transformer_decoder.layers]

    def _prepare(self):
        _module_bank(self).prepare(self.training)

    # ---- the two big linears ---------------------------------------------------------
    def _encode(self, x):
        B, S = x.shape[:2]
        flat = x.contiguous().reshape(B * S, -1)
        return ops.BigLinearFn.apply(flat, self.stft_to_embedding.weight, self.stft_to_embedding.bias).view(B, S, self.d_model)

    def encode_input(self, x):
        """SimpleDecoder_TransformerOnly.py:56-60: (B,S,2,287,513) -> (B,S,d_model)."""
        return self._encode(x)

    def _generate(self, tok):
        B, S, D = tok.shape
        h = L.layer_norm(tok, self.output_norm).reshape(B * S, D)
        return ops.BigLinearFn.apply(h, self.embedding_to_stft.weight, self.embedding_to_stft.bias).view(B, S, 2, 287, 513)

    def generate_output(self, decoder_outputs):
        """SimpleDecoder_TransformerOnly.py:62-66."""
        return self._generate(decoder_outputs)

    def create_causal_mask(self, seq_len, device=None):
        return torch.triu(torch.ones(seq_len, seq_len, device=device), diagonal=1).bool()

    def _memory(self, content_emb, class_emb):
        B, Sc, D = content_emb.shape
        cm = L.linear(content_emb.reshape(B * Sc, D), self._cp).view(B, Sc, D)
        km = L.linear(class_emb, self._kp).unsqueeze(1).expand(-1, Sc, -1)
        return ops.dropout(torch.cat([cm, km], dim=1), self.dropout.p, self.training)

    def prepare_memory(self, content_emb, class_emb):
        """SimpleDecoder_TransformerOnly.py:72-78."""
        self._prepare()
        return self._memory(content_emb, class_emb)

    def _stack(self, tgt, memory):
        for lyr in self._layers:
            tgt = lyr(tgt, memory, self.training)
        return tgt

    def _training_pass(self, y, memory):
        B = y.shape[0]
        emb = self._encode(y)
        tgt = torch.cat([self.start_token.expand(B, 1, -1), emb[:, :-1, :]], dim=1)
        tgt = L.layer_norm(self.pos_encoding(tgt), self.input_norm)
        return self._generate(self._stack(tgt, memory))

    def forward_training(self, y, memory):
        """SimpleDecoder_TransformerOnly.py:80-102."""
        self._prepare()
        return self._training_pass(y, memory)

    def _inference_pass(self, memory, target_length=None):
        B = memory.size(0)
        S = memory.size(1) // 2 if target_length is None else target_length
        seq = self.start_token.expand(B, -1, -1)
        outs = []
        for _ in range(S):
            nxt = self._stack(self.pos_encoding(seq), memory)[:, -1:, :]
            outs.append(nxt)
            seq = torch.cat([seq, nxt], dim=1)
        return self._generate(torch.cat(outs, dim=1))

    def forward_inference(self, memory, target_length=None):
        """SimpleDecoder_TransformerOnly.py:104-125."""
        self._prepare()
        return self._inference_pass(memory, target_length)

    def forward(self, content_emb, class_emb, y=None, target_length=None):
        """SimpleDecoder_TransformerOnly.py:127-133."""
        self._prepare()
        memory = self._memory(content_emb, class_emb)
        if self.training and y is not None:
            return self._training_pass(y, memory)
        return self._inference_pass(memory, target_length)


def compute_comprehensive_loss(output, target, lambda_temporal=0.3, lambda_phase=0.2, lambda_spectral=0.1):
    """SimpleDecoder_TransformerOnly.py:136-204: new_decoder's loss with the MSE term weighted 1.0 (:194) instead of
    2.0 (new_decoder.py:406); same single-pass kernel."""
    return _comprehensive_loss(output, target, lambda_temporal, lambda_phase, lambda_spectral, 1.0)

"""state_dict layouts (key -> empty tensor of the right shape/dtype) of the four
reference models, written out by hand so the oracle can be parameterised without
the reference on the path.  TEST INFRASTRUCTURE ONLY.

Pinned by the `digest_*` fields of tests/golden/model_*.npz, which were computed
from the real reference modules' state_dict() (key names and
shapes are part of the drop-in contract, incl. weight_orig/_u/_v and BN buffers).
"""
from __future__ import annotations

import torch

from .ast_oracle import positional_encoding


def _f(*shape):
    return torch.zeros(*shape, dtype=torch.float32)


def _sn_conv(d, p, cout, cin, k, transposed=False):
    d[p + "bias"] = _f(cout)
    if transposed:   # ConvTranspose2d weight is (in, out, k, k); spectral_norm dim=1
        d[p + "weight_orig"] = _f(cin, cout, k, k)
        d[p + "weight_u"] = _f(cout)
        d[p + "weight_v"] = _f(cin * k * k)
    else:
        d[p + "weight_orig"] = _f(cout, cin, k, k)
        d[p + "weight_u"] = _f(cout)
        d[p + "weight_v"] = _f(cin * k * k)


def _bn(d, p, c):
    d[p + "weight"] = _f(c)
    d[p + "bias"] = _f(c)
    d[p + "running_mean"] = _f(c)
    d[p + "running_var"] = _f(c)
    d[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def _affine(d, p, c):
    d[p + "weight"] = _f(c)
    d[p + "bias"] = _f(c)


def _linear(d, p, out_f, in_f):
    d[p + "weight"] = _f(out_f, in_f)
    d[p + "bias"] = _f(out_f)


def _mha(d, p, dm):
    d[p + "in_proj_weight"] = _f(3 * dm, dm)
    d[p + "in_proj_bias"] = _f(3 * dm)
    _linear(d, p + "out_proj.", dm, dm)


def _resblocks(d, p, chans=(32, 64, 128, 256, 512, 512), cin=2):
    for i, c in enumerate(chans):
        b = f"{p}{i}."
        _sn_conv(d, b + "conv1.", c, cin, 3)
        _bn(d, b + "bn1.", c)
        _sn_conv(d, b + "conv2.", c, c, 3)
        _bn(d, b + "bn2.", c)
        _sn_conv(d, b + "downsample.0.", c, cin, 1)
        _affine(d, b + "downsample.1.", c)
        cin = c


def _encoder_layers(d, p, n=4, dm=256, ff=1024):
    for i in range(n):
        b = f"{p}{i}."
        _mha(d, b + "self_attn.", dm)
        _linear(d, b + "linear1.", ff, dm)
        _linear(d, b + "linear2.", dm, ff)
        _affine(d, b + "norm1.", dm)
        _affine(d, b + "norm2.", dm)


def style_encoder_layout():
    d = {"cls_token": _f(1, 1, 256)}
    _resblocks(d, "cnn.net.")
    _linear(d, "cnn.proj.", 256, 512)
    d["pos_encoder.pe"] = positional_encoding(500, 256)[None]
    _affine(d, "norm.", 256)
    _encoder_layers(d, "transformer.layers.")
    return d


def content_encoder_layout():
    d = {}
    _resblocks(d, "cnn.")
    _linear(d, "proj.", 256, 512)
    d["pos_encoder.pe"] = positional_encoding(500, 256)[None]
    _affine(d, "norm.", 256)
    _encoder_layers(d, "transformer.layers.")
    return d


def decoder_layout():
    d = {"start_token": _f(1, 1, 256)}
    for idx, (cin, cout) in zip((0, 3, 6, 9), ((2, 16), (16, 32), (32, 64), (64, 64))):
        _sn_conv(d, f"conv_encoder.{idx}.", cout, cin, 3)
        _bn(d, f"conv_encoder.{idx + 1}.", cout)
    _sn_conv(d, "spatial_projection.0.", 64, 64, 3)
    _bn(d, "spatial_projection.1.", 64)
    _sn_conv(d, "spatial_projection.3.", 1, 64, 1)
    _linear(d, "feature_to_sequence.", 256, 512)
    _linear(d, "sequence_to_feature.", 512, 256)
    for idx, (cin, cout) in zip((0, 3, 6, 9), ((1, 64), (64, 32), (32, 16), (16, 8))):
        _sn_conv(d, f"conv_decoder.{idx}.", cout, cin, 3, transposed=True)
        _bn(d, f"conv_decoder.{idx + 1}.", cout)
    _sn_conv(d, "conv_decoder.12.", 2, 8, 3, transposed=True)
    _linear(d, "content_proj.", 256, 256)
    _linear(d, "class_proj.", 256, 256)
    d["pos_encoding.pe"] = positional_encoding(500, 256)[None]
    for i in range(4):
        b = f"transformer_decoder.layers.{i}."
        _mha(d, b + "self_attn.", 256)
        _mha(d, b + "multihead_attn.", 256)
        _linear(d, b + "linear1.", 1024, 256)
        _linear(d, b + "linear2.", 256, 1024)
        for n in ("norm1.", "norm2.", "norm3."):
            _affine(d, b + n, 256)
    _affine(d, "input_norm.", 256)
    _affine(d, "output_norm.", 256)
    return d


def simple_decoder_layout():
    """SimpleDecoder_TransformerOnly.py:9-44 (registration order of the reference)."""
    d = {"start_token": _f(1, 1, 256)}
    _linear(d, "stft_to_embedding.", 256, 2 * 287 * 513)
    _linear(d, "embedding_to_stft.", 2 * 287 * 513, 256)
    _linear(d, "content_proj.", 256, 256)
    _linear(d, "class_proj.", 256, 256)
    d["pos_encoding.pe"] = positional_encoding(500, 256)[None]
    for i in range(4):
        b = f"transformer_decoder.layers.{i}."
        _mha(d, b + "self_attn.", 256)
        _mha(d, b + "multihead_attn.", 256)
        _linear(d, b + "linear1.", 1024, 256)
        _linear(d, b + "linear2.", 256, 1024)
        for n in ("norm1.", "norm2.", "norm3."):
            _affine(d, b + n, 256)
    _affine(d, "input_norm.", 256)
    _affine(d, "output_norm.", 256)
    return d


def discriminator_layout():
    d = {}
    _linear(d, "net.0.", 128, 256)
    _linear(d, "net.2.", 128, 128)
    _linear(d, "net.4.", 2, 128)
    return d


LAYOUTS = {"style": style_encoder_layout, "content": content_encoder_layout,
           "decoder": decoder_layout, "disc": discriminator_layout, "simple_decoder": simple_decoder_layout}


def seeded_model_state(tag: str, requires_grad: bool = True):
    """Seeded parameters for one model as an oracle state dict."""
    from .seeded_params import seeded_state_dict
    sd = seeded_state_dict(LAYOUTS[tag](), tag=tag)
    if requires_grad:
        buffers = ("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked", "pe")
        for k, v in sd.items():
            if not k.endswith(buffers):
                v.requires_grad_(True)
    return sd

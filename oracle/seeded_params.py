"""Deterministic, order-independent parameter recipe shared by the golden
generator (tools/make_golden.py, which loads it INTO the reference modules) and
the tests (which load it into the oracle and into the HIP-backed modules).

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported by the product
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may use it.

Why a recipe instead of a saved state_dict: the three models hold 31 M fp32
parameters (124 MB) -- far too large for a fixture -- and a freshly constructed
reference decoder outputs exactly 0 (new_decoder.py:134-143 zeroes every 1-D
`*weight*`), which would make a parity check vacuous.  Every tensor
is drawn from its own PCG64 stream keyed by (seed, crc32(key)), so the values do
not depend on state_dict ordering or on which module family is being filled.
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

SEED = 1234


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode("utf-8"))])


def _fans(shape):
    if len(shape) < 2:
        return shape[0], shape[0]
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


def seeded_tensor(key: str, ref: torch.Tensor, seed: int = SEED) -> torch.Tensor | None:
    """Value for one state_dict entry, or None to keep the constructed value
    (sinusoidal `pe` tables, which are a pure function of the shape)."""
    shape = tuple(ref.shape)
    g = _rng(seed, key)
    leaf = key.rsplit("/", 1)[-1].rsplit(".", 1)[-1]

    if leaf == "pe":
        return None
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if leaf in ("weight_u", "weight_v"):
        v = g.standard_normal(shape).astype(np.float64)
        v /= max(np.linalg.norm(v), 1e-12)
        return torch.from_numpy(v.astype(np.float32))
    if leaf == "running_mean":
        return torch.from_numpy((0.1 * g.standard_normal(shape)).astype(np.float32))
    if leaf == "running_var":
        return torch.from_numpy(g.uniform(0.5, 1.5, shape).astype(np.float32))
    if leaf in ("cls_token", "start_token"):
        return torch.from_numpy(g.standard_normal(shape).astype(np.float32))
    if leaf == "weight_orig":
        fan_in, _ = _fans(shape)
        std = math.sqrt(2.0 / fan_in)
        return torch.from_numpy((std * g.standard_normal(shape)).astype(np.float32))
    if leaf in ("weight", "in_proj_weight"):
        if len(shape) == 1:  # BatchNorm / InstanceNorm / LayerNorm gamma
            return torch.from_numpy(g.uniform(0.5, 1.5, shape).astype(np.float32))
        fan_in, fan_out = _fans(shape)
        std = math.sqrt(2.0 / (fan_in + fan_out))
        return torch.from_numpy((std * g.standard_normal(shape)).astype(np.float32))
    if leaf in ("bias", "in_proj_bias"):
        return torch.from_numpy((0.05 * g.standard_normal(shape)).astype(np.float32))
    raise KeyError(f"seeded_params: no rule for state_dict key {key!r} shape {shape}")


def seeded_state_dict(template: dict, seed: int = SEED, tag: str = "") -> dict:
    """Return a new dict with every entry of `template` replaced by its seeded
    value.  `tag` namespaces the streams per model ("style", "content", ...) so
    the two encoders do not share weights."""
    out = {}
    for k in sorted(template.keys()):
        t = seeded_tensor(f"{tag}/{k}" if tag else k, template[k], seed)
        out[k] = template[k].detach().clone() if t is None else t.reshape(template[k].shape)
    return out


def layout_digest(template: dict) -> str:
    """Hash of (key, shape, dtype) over a state_dict -- the golden file stores it
    so a silent state_dict layout drift fails the fixture check."""
    h = zlib.crc32(b"")
    for k in sorted(template.keys()):
        v = template[k]
        h = zlib.crc32(f"{k}:{tuple(v.shape)}:{v.dtype};".encode(), h)
    return f"{h:08x}"


def seeded_input(B: int, S: int, seed: int = 1000, F: int = 597) -> torch.Tensor:
    """x ~ N(0,1) of shape (B,S,2,287,F): what the reference's own smoke tests
    feed the encoders (test_correctness.ipynb cell 6)."""
    g = np.random.default_rng([seed, B, S, F])
    return torch.from_numpy(g.standard_normal((B, S, 2, 287, F)).astype(np.float32))


def seeded_normal(shape, seed: int) -> torch.Tensor:
    """N(0,1) tensor of any shape from its own PCG64 stream (embedding-shaped test inputs)."""
    g = np.random.default_rng([seed] + list(shape))
    return torch.from_numpy(g.standard_normal(tuple(shape)).astype(np.float32))


def balanced_labels(B: int) -> torch.Tensor:
    """dataloader.py:143-146 -- first half piano (0), second half violin (1)."""
    return torch.cat([torch.zeros(B // 2, dtype=torch.long), torch.ones(B - B // 2, dtype=torch.long)])

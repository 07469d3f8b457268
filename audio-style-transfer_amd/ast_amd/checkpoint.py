"""Checkpoint interop and dataset statistics.

* `save_checkpoint` / `load_checkpoint`: the reference's `.pth` layout -- one dict with the four state_dicts under
  'content_encoder', 'style_encoder', 'decoder', 'discriminator' (evaluation_style_transfer.py:246-252).  The modules of
  this package keep the reference's parameter/buffer names (spectral-norm `weight_orig/_u/_v`, BatchNorm buffers,
  `pos_encoder.pe`, ...), so checkpoints move both ways without key mapping.
* `StftStats`: Preprocessing_Dataset/compute_unified_stats.py for the STFT bins, as a streaming device reduction
  (mean over clips of the per-clip mean; sqrt of the mean per-clip unbiased variance).  The CQT half of that script is
  librosa arithmetic (parity-unpinned) and is not built."""
from __future__ import annotations

import numpy as np
import torch

from . import utilityFunctions as U
from ._lib import check, lib, ptr, stream

KEYS = ("content_encoder", "style_encoder", "decoder", "discriminator")


def save_checkpoint(path, content_encoder, style_encoder, decoder, discriminator, **extra):
    state = {"content_encoder": content_encoder.state_dict(), "style_encoder": style_encoder.state_dict(),
             "decoder": decoder.state_dict(), "discriminator": discriminator.state_dict()}
    state.update(extra)
    torch.save(state, path)


def load_checkpoint(path, content_encoder=None, style_encoder=None, decoder=None, discriminator=None, map_location="cuda", strict=True):
    """Loads whichever modules are given; returns the raw checkpoint dict (extra entries included)."""
    ckpt = torch.load(path, map_location=map_location)
    for key, mod in zip(KEYS, (content_encoder, style_encoder, decoder, discriminator)):
        if mod is not None:
            mod.load_state_dict(ckpt[key], strict=strict)
    return ckpt


class StftStats:
    """stats = StftStats(); for wave in clips: stats.add(wave); mean, std = stats.finalize()   ((2,513) each)."""

    def __init__(self, device="cuda", n_bins=513):
        self.mean_acc = torch.zeros(2, n_bins, dtype=torch.float32, device=device)
        self.var_acc = torch.zeros_like(self.mean_acc)
        self.count = 0

    def add(self, waveform):
        spec = U.get_STFT(waveform.to(self.mean_acc.device)).contiguous()          # (2, T, 513)
        C_, T, F = spec.shape
        check(lib().ast_bin_stats_acc(ptr(spec), ptr(self.mean_acc), ptr(self.var_acc), C_, T, F, stream()), "ast_bin_stats_acc")
        self.count += 1

    def finalize(self):
        if self.count == 0:
            raise ValueError("StftStats.finalize: no clips were added")
        return self.mean_acc / self.count, torch.sqrt(self.var_acc / self.count)

    def save(self, path, cqt_mean=None, cqt_std=None):
        """train_set_stats/*.npz layout (stft_mean, stft_std[, cqt_mean, cqt_std])."""
        mean, std = self.finalize()
        out = {"stft_mean": mean.cpu().numpy(), "stft_std": std.cpu().numpy()}
        if cqt_mean is not None:
            out.update(cqt_mean=np.asarray(cqt_mean), cqt_std=np.asarray(cqt_std))
        np.savez(path, **out)

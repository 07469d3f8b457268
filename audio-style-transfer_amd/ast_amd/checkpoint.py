"""Checkpoint interop and dataset statistics.

* `save_checkpoint` / `load_checkpoint`: the reference's `.pth` layout -- one dict with the four state_dicts under
  'content_encoder', 'style_encoder', 'decoder', 'discriminator' (evaluation_style_transfer.py:246-252).  The modules of
  this package keep the reference's parameter/buffer names (spectral-norm `weight_orig/_u/_v`, BatchNorm buffers,
  `pos_encoder.pe`, ...), so checkpoints move both ways without key mapping.
* `UnifiedStats`: Preprocessing_Dataset/compute_unified_stats.py:25-68 -- per-bin statistics over the STFT || CQT
  concatenation (2, T, 513 + 84) of every clip -- as a streaming device reduction (mean over clips of the per-clip mean; sqrt
  of the mean per-clip unbiased variance), saved in the train_set_stats/*.npz layout.  The CQT bins come from the device
  CQT of cqt.py (librosa's algorithm restated: parity unpinned, as everywhere else).  `StftStats` is the STFT-only form."""
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
        """(STFT-only accumulator; UnifiedStats.save writes all four arrays from its own data.)"""
        return self._save(path, cqt_mean, cqt_std)

    def _save(self, path, cqt_mean=None, cqt_std=None):
        """train_set_stats/*.npz layout (stft_mean, stft_std[, cqt_mean, cqt_std])."""
        mean, std = self.finalize()
        out = {"stft_mean": mean.cpu().numpy(), "stft_std": std.cpu().numpy()}
        if cqt_mean is not None:
            out.update(cqt_mean=np.asarray(cqt_mean), cqt_std=np.asarray(cqt_std))
        np.savez(path, **out)


class UnifiedStats:
    """compute_unified_stats.py:25-68: stats = UnifiedStats(); for wave in clips: stats.add(wave); stats.save(path).

    add() takes one mono clip (samples,) or (1, samples); the clip's STFT (get_STFT) and CQT (get_CQT) are concatenated on the
    bin axis as the script's concat_stft_cqt does and the per-clip mean / unbiased variance over time of all 597 bins are
    accumulated by one kernel launch (ast_bin_stats_acc)."""

    N_STFT, N_CQT = 513, 84

    def __init__(self, device="cuda"):
        F = self.N_STFT + self.N_CQT
        self.mean_acc = torch.zeros(2, F, dtype=torch.float32, device=device)
        self.var_acc = torch.zeros_like(self.mean_acc)
        self.count = 0

    def add(self, waveform):
        w = waveform.to(self.mean_acc.device)
        merged = torch.cat((U.get_STFT(w), U.get_CQT(w).to(self.mean_acc.device)), dim=2).contiguous()      # (2, T, 597)
        C_, T, F = merged.shape
        check(lib().ast_bin_stats_acc(ptr(merged), ptr(self.mean_acc), ptr(self.var_acc), C_, T, F, stream()), "ast_bin_stats_acc")
        self.count += 1

    def finalize(self):
        """(stft_mean, stft_std, cqt_mean, cqt_std), each (2, bins)."""
        if self.count == 0:
            raise ValueError("UnifiedStats.finalize: no clips were added")
        mean, std = self.mean_acc / self.count, torch.sqrt(self.var_acc / self.count)
        n = self.N_STFT
        return mean[:, :n], std[:, :n], mean[:, n:], std[:, n:]

    def save(self, path):
        """train_set_stats/stats_unified_stft_cqt.npz layout (compute_unified_stats.py:62-68)."""
        sm, ss, cm, cs = (t.cpu().numpy() for t in self.finalize())
        np.savez(path, stft_mean=sm, stft_std=ss, cqt_mean=cm, cqt_std=cs)

"""Drop-in for the torch-only part of the reference's utilityFunctions.py on libast_hip.

get_STFT runs the LDS-staged radix-4 FFT kernel; get_overlap_windows /
sections2spectrogram / concat_stft_cqt are index plumbing on device tensors.
inverse_STFT runs an inverse-FFT + overlap-add kernel pair.  get_CQT and load_audio live in cqt.py (librosa /
torchaudio arithmetic restated from their published algorithms: parity unpinned); inverse_CQT is not built.
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, stream

WINDOW_SIZE = 287
OVERLAP_PERCENTAGE = 0.3
OVERLAP_FRAMES = 96


def _ident_stats(device):
    return torch.zeros(2, 513, device=device), torch.ones(2, 513, device=device) - 1e-8


def stft_sections(waves: torch.Tensor, mean=None, std=None, n_sections=None, F_total=513, out=None):
    """waves (Bc, n) f32 cuda -> (Bc, S, 2, 287, F_total): STFT + z-score + windowing in one kernel
    (utilityFunctions.py:12-37,240-263; dataloader.py:9-13).  Bins >= 513 are left untouched."""
    Bc, n = waves.shape
    T = 1 + n // 256
    step = WINDOW_SIZE - OVERLAP_FRAMES
    if n_sections is None:
        n_sections = len(section_starts(T))
    if mean is None:
        mean, std = _ident_stats(waves.device)
    if out is None:
        out = torch.zeros((Bc, n_sections, 2, WINDOW_SIZE, F_total), dtype=torch.float32, device=waves.device)
    check(lib().ast_stft_sections(ptr(waves.contiguous()), Bc, n, ptr(mean.contiguous()), ptr(std.contiguous()), ptr(out),
                                  n_sections, WINDOW_SIZE, step, F_total, stream()), "ast_stft_sections")
    # version bump: the kernel wrote through the raw pointer (keeps version-keyed caches honest).  NOT `out.add_(0)`: that is a
    # read-modify-write of ALL bins on this stream, and the CQT kernel fills bins >= 513 of the same tensor on another stream
    # (train.Trainer._run_frontend) -- the no-op add wrote stale CQT bins back now and then (found by the mixed-length oracle test)
    torch.autograd.graph.increment_version(out)
    return out


def section_starts(n_time, window_size=WINDOW_SIZE, overlap_frames=OVERLAP_FRAMES):
    step, starts = window_size - overlap_frames, []
    for s in range(0, n_time, step):
        e = min(s + window_size, n_time)
        if e - s < window_size * 0.5:
            break
        starts.append(s)
        if e == n_time:
            break
    return starts


def get_STFT(waveform, n_fft=1024, hop_length=256):
    """utilityFunctions.py:12-37: (channels, samples) or (samples,) -> (2, T, 513)."""
    if n_fft != 1024 or hop_length != 256:
        raise NotImplementedError("the HIP front-end is built for n_fft=1024, hop=256 (the reference's only configuration)")
    w = waveform.reshape(1, -1).float()
    n = w.shape[1]
    T = 1 + n // 256
    # one "section" of T rows, step irrelevant: reuse the section kernel with win = T
    mean, std = _ident_stats(w.device)
    out = torch.empty((1, 1, 2, T, 513), dtype=torch.float32, device=w.device)
    check(lib().ast_stft_sections(ptr(w.contiguous()), 1, n, ptr(mean), ptr(std), ptr(out), 1, T, T, 513, stream()),
          "ast_stft_sections")
    return out[0, 0]


def get_overlap_windows(spectrogram, window_size=WINDOW_SIZE, overlap_frames=OVERLAP_FRAMES):
    """utilityFunctions.py:240-263: (2,T,F) -> (S,2,window,F); tails shorter than half a window are dropped."""
    Cc, n_time, n_freq = spectrogram.shape
    starts = section_starts(n_time, window_size, overlap_frames)
    out = spectrogram.new_zeros((len(starts), Cc, window_size, n_freq))
    for i, s in enumerate(starts):
        e = min(s + window_size, n_time)
        out[i, :, :e - s] = spectrogram[:, s:e]
    return out


def sections2spectrogram(sections, original_size, overlap=OVERLAP_FRAMES):
    """utilityFunctions.py:265-283: count-normalised overlap-average, (S,2,wind,F) -> (2,original_size,F), as one
    HIP kernel (ast_sections_overlap_avg)."""
    return sections2spectrogram_batch(sections.unsqueeze(0), original_size, overlap)[0]


def sections2spectrogram_batch(sections, original_size, overlap=OVERLAP_FRAMES, n_bins=None):
    """Batched form: (B,S,2,wind,F) -> (B,2,original_size,n_bins or F)."""
    B, S, _, wind, F = sections.shape
    hop = wind - overlap
    n_bins = F if n_bins is None else n_bins
    out_T = min(int(original_size), hop * (S - 1) + wind)
    sec = sections.float().contiguous()
    out = torch.empty((B, 2, out_T, n_bins), dtype=torch.float32, device=sec.device)
    check(lib().ast_sections_overlap_avg(ptr(sec), ptr(out), B, S, wind, hop, F, n_bins, out_T, stream()), "ast_sections_overlap_avg")
    return out


def concat_stft_cqt(stft, cqt):
    """utilityFunctions.py:285-299."""
    if stft.ndim != 3 or cqt.ndim != 3:
        raise ValueError(f"Both tensors must be 3D, got {stft.ndim}D e {cqt.ndim}D.")
    if stft.shape[0] != cqt.shape[0] or stft.shape[1] != cqt.shape[1]:
        raise ValueError(f"Channel/Time mismatch: stft {stft.shape[:2]} vs cqt {cqt.shape[:2]}")
    return torch.cat([stft, cqt], dim=2)


def inverse_STFT(stft_tensor, n_fft=1024, hop_length=256):
    """utilityFunctions.py:62-82: (2, T, 513) -> waveform (256*(T-1),) via the HIP inverse-FFT + overlap-add kernels."""
    if n_fft != 1024 or hop_length != 256:
        raise NotImplementedError("the HIP front-end is built for n_fft=1024, hop=256 (the reference's only configuration)")
    return inverse_STFT_batch(stft_tensor.unsqueeze(0))[0]


def inverse_STFT_batch(spec):
    """(B, 2, T, 513) -> (B, 256*(T-1)) waveforms in one launch pair."""
    spec = spec.float().contiguous()
    B, _, T, _ = spec.shape
    frames = torch.empty((B, T, 1024), dtype=torch.float32, device=spec.device)
    wave = torch.empty((B, 256 * (T - 1)), dtype=torch.float32, device=spec.device)
    check(lib().ast_istft(ptr(spec), B, T, ptr(frames), ptr(wave), stream()), "ast_istft")
    return wave


from .cqt import get_CQT, load_audio, resample, cqt_batch, cqt_sections  # noqa: E402,F401  (utilityFunctions.py:39-60,105-122)


def inverse_CQT(*a, **k):
    raise NotImplementedError("inverse_CQT (librosa.icqt, utilityFunctions.py:84-103) is off the train/inference path and not built")

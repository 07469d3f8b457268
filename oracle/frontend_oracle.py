"""CPU oracle for the signal front-end (utilityFunctions.py / dataloader.py).

TEST INFRASTRUCTURE ONLY -- see oracle/ast_oracle.py for the import rule.

STFT, windowing, overlap-average and iSTFT are written as explicit framed DFTs
in numpy float64->float32 and are pinned by tests/golden/frontend.npz (generated
by running the reference's own get_STFT / get_overlap_windows /
sections2spectrogram / inverse_STFT in the build container).

CQT: PARITY UNPINNED.  The arithmetic lives in librosa (unpinned version,
README.md:165), which is not installed anywhere this project runs; the
reference's tests pin only the output shape (2,T,84).  No CQT oracle is
provided in this round.
"""
from __future__ import annotations

import numpy as np

N_FFT = 1024
HOP = 256
WINDOW_SIZE = 287      # utilityFunctions.py:8
OVERLAP_FRAMES = 96    # utilityFunctions.py:10


def hann_periodic(n=N_FFT):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float64)


def stft(wave: np.ndarray, n_fft=N_FFT, hop=HOP) -> np.ndarray:
    """utilityFunctions.py:12-37 -> (2, T, n_fft/2+1) float32.  torch.stft
    defaults: center=True, reflect padding n_fft/2, periodic Hann, onesided,
    unnormalised; T = 1 + len//hop."""
    w = np.asarray(wave, dtype=np.float64).reshape(-1)
    pad = n_fft // 2
    wp = np.pad(w, (pad, pad), mode="reflect")
    T = 1 + (len(wp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    frames = wp[idx] * hann_periodic(n_fft)[None, :]
    spec = np.fft.rfft(frames, axis=1)               # (T, 513)
    return np.stack([spec.real, spec.imag], 0).astype(np.float32)


def istft(spec: np.ndarray, n_fft=N_FFT, hop=HOP) -> np.ndarray:
    """utilityFunctions.py:62-82 (torch.istft defaults: center=True, length=None):
    windowed inverse frames overlap-added, divided by the overlap-added window^2
    envelope, n_fft/2 trimmed from both ends."""
    z = spec[0].astype(np.float64) + 1j * spec[1].astype(np.float64)   # (T, 513)
    T = z.shape[0]
    win = hann_periodic(n_fft)
    frames = np.fft.irfft(z, n=n_fft, axis=1) * win[None, :]
    L = n_fft + hop * (T - 1)
    y = np.zeros(L)
    env = np.zeros(L)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += frames[t]
        env[t * hop:t * hop + n_fft] += win ** 2
    pad = n_fft // 2
    y, env = y[pad:L - pad], env[pad:L - pad]
    return (y / np.where(env > 1e-11, env, 1.0)).astype(np.float32)


def normalize(x: np.ndarray, mean: np.ndarray, std: np.ndarray, eps=1e-8) -> np.ndarray:
    """dataloader.py:9-13: eps is added to std, not var."""
    return ((x - mean[:, None, :]) / (std[:, None, :] + eps)).astype(np.float32)


def section_starts(n_time, window=WINDOW_SIZE, overlap=OVERLAP_FRAMES):
    """utilityFunctions.py:246-262: step = window-overlap; a tail shorter than
    half a window is dropped, otherwise zero padded."""
    step = window - overlap
    starts = []
    for s in range(0, n_time, step):
        e = min(s + window, n_time)
        if e - s < window * 0.5:
            break
        starts.append(s)
        if e == n_time:
            break
    return starts


def overlap_windows(spec: np.ndarray, window=WINDOW_SIZE, overlap=OVERLAP_FRAMES) -> np.ndarray:
    """(2,T,F) -> (S,2,window,F)."""
    starts = section_starts(spec.shape[1], window, overlap)
    out = np.zeros((len(starts), spec.shape[0], window, spec.shape[2]), dtype=spec.dtype)
    for i, s in enumerate(starts):
        e = min(s + window, spec.shape[1])
        out[i, :, :e - s] = spec[:, s:e]
    return out


def sections_to_spectrogram(sections: np.ndarray, original_size: int, overlap=OVERLAP_FRAMES) -> np.ndarray:
    """utilityFunctions.py:265-283: overlap-AVERAGE (count normalised)."""
    S, C, Wn, Fq = sections.shape
    hop = Wn - overlap
    n_time = hop * (S - 1) + Wn
    full = np.zeros((C, n_time, Fq), dtype=np.float64)
    cnt = np.zeros((1, n_time, 1))
    for i in range(S):
        full[:, i * hop:i * hop + Wn] += sections[i]
        cnt[:, i * hop:i * hop + Wn] += 1.0
    return (full / np.maximum(cnt, 1.0))[:, :original_size].astype(np.float32)


def collate(piano_sections, violin_sections):
    """dataloader.py:123-147 with `batch` already reduced to the half that is
    used: rows [piano_0..piano_{h-1}, violin_0..violin_{h-1}], labels [0]*h+[1]*h."""
    x = np.stack(list(piano_sections) + list(violin_sections), 0)
    h = len(piano_sections)
    labels = np.array([0] * h + [1] * len(violin_sections), dtype=np.int64)
    return x, labels


def synth_waveform(i: int, kind: str, seconds: float = 4.0, sr: int = 22050) -> np.ndarray:
    """SURVEY 8(d) synthetic clips: piano-like = decaying harmonic partials of a
    random MIDI note re-struck every 0.5 s; violin-like = 8 harmonics with 5.5 Hz
    vibrato and a slow envelope; -40 dB noise; RMS 0.07."""
    g = np.random.default_rng(1000 + i)
    n = int(round(seconds * sr))
    t = np.arange(n) / sr
    y = np.zeros(n)
    if kind == "piano":
        for k in range(int(np.ceil(seconds / 0.5))):
            f0 = 440.0 * 2 ** ((g.integers(40, 89) - 69) / 12)
            tt = t - 0.5 * k
            on = tt >= 0
            for h in range(1, 7):
                y += on * np.exp(-3.0 * h * np.clip(tt, 0, None)) * np.sin(2 * np.pi * f0 * h * tt) / h
    else:
        f0 = 440.0 * 2 ** ((g.integers(55, 89) - 69) / 12)
        vib = 0.01 * np.sin(2 * np.pi * 5.5 * t)
        env = np.minimum(1.0, t / 0.3) * np.minimum(1.0, (seconds - t) / 0.3)
        for h in range(1, 9):
            y += env * np.sin(2 * np.pi * f0 * h * (t + vib / (2 * np.pi * 5.5) * 5.5)) / h
    y += 10 ** (-40 / 20) * g.standard_normal(n)
    y *= 0.07 / max(np.sqrt(np.mean(y ** 2)), 1e-9)
    return y.astype(np.float32)

"""get_CQT on the device (utilityFunctions.py:39-60: `librosa.cqt(y, sr=22050, n_bins=84, hop_length=256)`) and
load_audio (utilityFunctions.py:105-122).

PARITY UNPINNED: librosa and torchaudio are absent from every image the build sees and the reference pins only the
CQT's output shape, so what is built here follows the *published* algorithms (librosa >= 0.10 `vqt` with its `cqt`
defaults: fmin = C1, 12 bins per octave, filter_scale 1, norm 1, sparsity 0.01, hann wavelets, scale=True,
pad_mode "constant"; torchaudio's `sinc_interp_hann` resampler with lowpass_filter_width 6, rolloff 0.99), and is
checked against the independent restatement in oracle/cqt_oracle.py and against known answers (a sinusoid at a bin's
centre frequency).  One piece cannot be restated: librosa halves the signal between octaves with the closed soxr
library (`res_type="soxr_hq"`); a linear-phase Kaiser half-band FIR with soxr-HQ's published band edges (pass 0.913
of the new Nyquist, stop at the new Nyquist, ~120 dB) stands in for it.

Per octave, librosa computes `fft_basis @ stft(y_o, window="ones")`.  Both factors are linear in the frame, so they
fold into 12 complex time-domain kernels of nfft taps -- the same 12 for every octave, because the sample rate and the
wavelet frequencies halve together -- and an octave is one strided correlation launch (csrc/cqt.hip).  The kernels
are a constant table (like FFT twiddles), built once in float64 on the host.
"""
from __future__ import annotations

import ctypes
import functools
import math
import wave as _wave

import numpy as np
import torch

from ._lib import check, lib, ptr, stream

HANN_BANDWIDTH = 1.50018310546875          # librosa.filters.window_bandwidth("hann")
C1_HZ = 32.70319566257483                  # librosa.note_to_hz("C1"), cqt's default fmin


def _halfband_taps():
    """Linear-phase half-band low-pass standing in for soxr_hq: edges 0.913 * fs/4 and fs/4 (of the input rate)."""
    att, f_pass, f_stop = 120.0, 0.913 * 0.25, 0.25
    df = f_stop - f_pass
    n = int(math.ceil((att - 7.95) / (14.36 * df))) | 1
    beta = 0.1102 * (att - 8.7)
    k = np.arange(n, dtype=np.float64) - (n - 1) / 2
    fc = 0.5 * (f_pass + f_stop)
    h = 2 * fc * np.sinc(2 * fc * k) * np.kaiser(n, beta)
    return h / h.sum()


@functools.lru_cache(maxsize=None)
def _plan(sr: float, n_bins: int, hop: int, bpo: int = 12, fmin: float = C1_HZ, filter_scale: float = 1.0, sparsity: float = 0.01):
    """Constant tables of the transform: per-octave correlation kernels, per-bin scales, decimator taps."""
    n_oct = int(math.ceil(n_bins / bpo))
    n_filt = min(bpo, n_bins)
    if hop % (1 << (n_oct - 1)):
        raise ValueError(f"hop_length must be a positive integer multiple of 2^{n_oct - 1} for {n_oct}-octave CQT")   # librosa's check
    freqs = fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bpo)
    # librosa.filters._relative_bandwidth on an equal-tempered grid is constant
    alpha = (2.0 ** (2.0 / bpo) - 1) / (2.0 ** (2.0 / bpo) + 1)
    Q = filter_scale / alpha
    lengths = Q * sr / freqs
    if (freqs * (1 + 0.5 * HANN_BANDWIDTH / Q)).max() > sr / 2:
        raise ValueError("wavelet basis pass-band lies beyond Nyquist; reduce n_bins")
    top, ltop = freqs[-n_filt:], lengths[-n_filt:]
    nfft = int(2.0 ** math.ceil(math.log2(ltop.max())))
    basis = np.zeros((n_filt, nfft), dtype=np.complex128)
    for i, (ilen, f) in enumerate(zip(ltop, top)):
        idx = np.arange(-ilen // 2, ilen // 2, dtype=np.float64)
        sig = np.exp(2j * np.pi * f / sr * idx)
        m = len(sig)
        sig = sig * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(m) / m))          # periodic hann (scipy get_window fftbins=True)
        sig = sig / np.abs(sig).sum()                                           # norm=1
        lpad = (nfft - m) // 2
        basis[i, lpad:lpad + m] = sig
    basis *= ltop[:, None] / nfft
    fb = np.fft.fft(basis, axis=1)[:, :nfft // 2 + 1]
    # util.sparsify_rows(quantile=0.01): drop the smallest magnitudes holding < 1 % of each row's L1 mass
    mags = np.abs(fb)
    srt = np.sort(mags, axis=1)
    cum = np.cumsum(srt / mags.sum(axis=1, keepdims=True), axis=1)
    thr = srt[np.arange(n_filt), np.argmin(cum < sparsity, axis=1)]
    fb = np.where(mags >= thr[:, None], fb, 0)
    # fold the one-sided rectangular-window DFT in: W[k][n] = sum_f fb[k][f] exp(-2 pi i f n / nfft)
    f_idx, n_idx = np.arange(nfft // 2 + 1)[:, None], np.arange(nfft)[None, :]
    W = fb @ np.exp(-2j * np.pi * f_idx * n_idx / nfft)
    # per octave o (0 = top): bins, hop, scale = sqrt(sr/my_sr) / sqrt(length of the bin at the original rate)
    octaves = []
    end = n_bins
    for o in range(n_oct):
        lo = max(0, end - n_filt)
        nb = end - lo
        rows = slice(n_filt - nb, n_filt)                                       # a short last octave keeps the top filters
        octaves.append((lo, nb, rows, hop >> o, math.sqrt(2.0 ** o) / np.sqrt(lengths[lo:end])))
        end = lo
    return {"nfft": nfft, "W": W, "octaves": octaves, "taps": _halfband_taps(), "n_filt": n_filt}


@functools.lru_cache(maxsize=None)
def _device_plan(sr, n_bins, hop, device):
    p = _plan(float(sr), int(n_bins), int(hop))
    dev = torch.device(device)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    oc = p["octaves"]
    return {"nfft": p["nfft"], "w_re": f32(p["W"].real), "w_im": f32(p["W"].imag), "taps": f32(p["taps"][None, :]),
            "ntaps": len(p["taps"]), "octaves": oc, "los": [o[0] for o in oc], "nfs": [o[1] for o in oc],
            "row0s": [o[2].start for o in oc], "hops": [o[3] for o in oc],
            "scale": f32(np.concatenate([o[4] for o in reversed(oc)]))}               # by bin, low to high


def resample_poly(x: torch.Tensor, kern: torch.Tensor, orig: int, new: int, width: int, m: int, gain: float = 1.0):
    """(B, n) f32 cuda -> (B, m): y[i*new + p] = gain * sum_k kern[p][k] x[i*orig + k - width]."""
    B, n = x.shape
    y = torch.empty((B, m), dtype=torch.float32, device=x.device)
    check(lib().ast_resample_poly(ptr(x), B, n, ptr(kern), orig, new, kern.shape[1], width, ptr(y), m, gain, stream()), "ast_resample_poly")
    return y


def cqt_batch(waves: torch.Tensor, sample_rate=22050, n_bins=84, hop_length=256, out=None, bin0=0):
    """(B, n) f32 cuda -> (B, 2, T, n_bins) f32 (real plane, imaginary plane), T = 1 + n // hop_length.
    With `out` (B, 2, T, ld) the bins land at out[..., bin0:bin0+n_bins] (e.g. behind the 513 STFT bins)."""
    if waves.dim() != 2 or waves.dtype != torch.float32 or not waves.is_cuda:
        raise ValueError("cqt_batch: (B, n) float32 cuda waveforms expected")
    waves = waves.contiguous()
    B, n = waves.shape
    p = _device_plan(float(sample_rate), int(n_bins), int(hop_length), str(waves.device))
    T = 1 + n // hop_length
    if out is None:
        out = torch.empty((B, 2, T, n_bins), dtype=torch.float32, device=waves.device)
    elif out.shape[:3] != (B, 2, T) or not out.is_contiguous() or out.shape[3] < bin0 + n_bins:
        raise ValueError("cqt_batch: out must be contiguous (B, 2, T, >= bin0 + n_bins)")
    ld = out.shape[3]
    ys, half = [waves], (p["ntaps"] - 1) // 2
    for _ in range(len(p["octaves"]) - 1):
        # audio.resample(orig_sr=2, target_sr=1, scale=True): ceil(m/2) samples, times sqrt(2)
        m = ys[-1].shape[1]
        ys.append(resample_poly(ys[-1], p["taps"], 2, 1, half, (m + 1) // 2, math.sqrt(2.0)))
    no = len(ys)
    I = ctypes.c_int * no
    assert all(1 + y.shape[1] // h >= T for y, h in zip(ys, p["hops"]))           # librosa trims every octave to the shortest
    check(lib().ast_cqt_octaves((ctypes.c_void_p * no)(*[y.data_ptr() for y in ys]), I(*[y.shape[1] for y in ys]), I(*p["hops"]), I(*p["los"]),
                                I(*p["nfs"]), I(*p["row0s"]), no, B, ptr(p["w_re"]), ptr(p["w_im"]), ptr(p["scale"]), p["nfft"], ptr(out), T, ld,
                                bin0, stream()), "ast_cqt_octaves")
    return out


def cqt_sections(waves: torch.Tensor, x: torch.Tensor, mean=None, std=None, sample_rate=22050, n_bins=84, hop_length=256,
                 bin0=513, window_size=287, overlap_frames=96):
    """The CQT twin of utilityFunctions.stft_sections: waves (Bc, n) -> CQT -> z-score -> overlap windows, written into
    x[..., bin0:bin0+n_bins] of the collate-layout batch x (Bc, S, 2, 287, F_total) (dataloader.py:94-121)."""
    Bc, S = x.shape[0], x.shape[1]
    if not x.is_contiguous() or x.shape[2] != 2 or x.shape[3] != window_size or x.shape[4] < bin0 + n_bins or waves.shape[0] != Bc:
        raise ValueError("cqt_sections: x must be contiguous (Bc, S, 2, window, >= bin0 + n_bins)")
    c = cqt_batch(waves, sample_rate, n_bins, hop_length)
    if mean is None:
        mean = torch.zeros(2, n_bins, device=x.device)
        std = torch.ones(2, n_bins, device=x.device) - 1e-8
    check(lib().ast_cqt_sections(ptr(c), Bc, c.shape[2], n_bins, ptr(mean.contiguous()), ptr(std.contiguous()), ptr(x), S, window_size,
                                 window_size - overlap_frames, x.shape[4], bin0, stream()), "ast_cqt_sections")
    return x


def get_CQT(waveform, sample_rate=22050, n_bins=84, hop_length=256):
    """utilityFunctions.py:39-60: (channels, samples) or (samples,) -> (2, T, n_bins) f32 [real, imaginary].
    The reference returns a CPU tensor (it goes through numpy); here the result stays on the waveform's GPU
    (`concat_stft_cqt` moves either to the STFT's device)."""
    if not isinstance(waveform, torch.Tensor):
        waveform = torch.as_tensor(np.asarray(waveform), dtype=torch.float32)
    w = waveform.squeeze()
    if w.dim() != 1:
        raise ValueError(f"get_CQT: mono waveform expected, got shape {tuple(waveform.shape)}")     # librosa would transform each channel
    if not w.is_cuda:
        w = w.to("cuda")
    return cqt_batch(w.float()[None], sample_rate, n_bins, hop_length)[0]


@functools.lru_cache(maxsize=None)
def _sinc_bank(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional._get_sinc_resample_kernel, sinc_interp_hann (orig, new already divided by their gcd)."""
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * np.pi / lowpass_filter_width / 2) ** 2
    t = t * np.pi
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * (base / orig)
    return kern.astype(np.float32), width


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int):
    """torchaudio.functional.resample(waveform, orig_freq, new_freq) with its defaults, on the device."""
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Original frequency and desired frequency should be positive")
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kern, width = _sinc_bank(orig, new)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).float().contiguous()
    if not x.is_cuda:
        x = x.to("cuda")
    m = int(math.ceil(new * shape[-1] / orig))
    y = resample_poly(x, torch.from_numpy(kern).to(x.device), orig, new, width, m)
    return y.reshape(shape[:-1] + (m,))


def _decode_wav(path):
    """PCM WAV -> (channels, samples) f32 in [-1, 1) with torchaudio.load's integer scaling, and the file's rate."""
    with _wave.open(path, "rb") as f:
        ch, sw, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"load_audio: unsupported sample width {sw}")
    return torch.from_numpy(np.ascontiguousarray(a.reshape(-1, ch).T)), sr


def load_audio(file_path, sample_rate=22050, cut_time_seconds=10, device="cuda"):
    """utilityFunctions.py:105-122: decode, zero-pad / cut to cut_time_seconds at the FILE's rate, resample to
    sample_rate, average a stereo pair.  Decoding is host work (PCM WAV through the stdlib; the dataset is WAV --
    other containers need torchaudio, which no image has); everything after it runs on the device."""
    waveform, orig_sr = _decode_wav(file_path)
    cut = int(cut_time_seconds * orig_sr)
    if waveform.shape[-1] < cut:
        waveform = torch.cat([waveform, torch.zeros((waveform.shape[0], cut - waveform.shape[-1]))], dim=-1)
    waveform = waveform[:, :cut].to(device)
    if orig_sr != sample_rate:
        waveform = resample(waveform, orig_sr, sample_rate)
    if waveform.shape[0] == 2:
        waveform = torch.mean(waveform, dim=0, keepdim=True)
    return waveform, sample_rate

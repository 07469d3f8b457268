"""TEST INFRASTRUCTURE ONLY -- CPU restatement of get_CQT and load_audio's resampler.  PARITY UNPINNED.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does.

What it restates:
  * get_CQT, /root/reference/utilityFunctions.py:39-60 = `librosa.cqt(y, sr=22050, n_bins=84, hop_length=256)`,
    everything else at librosa's defaults.  librosa is a third-party dependency that is absent from /root/reference and
    from every image, and the reference does not pin its version (README.md:165); its own tests pin only the output
    shape (2, 862, 84) for a 10 s clip (test_correctness.ipynb cell 3).  This file follows the published algorithm of
    librosa >= 0.10 (`core.constantq.vqt` with gamma = 0) step by step, in the order librosa runs it, in float64:
      wavelet_lengths -> wavelet (hann, L1-normalised, centred in nfft) -> * lengths/nfft -> FFT -> sparsify_rows(0.01)
      per octave: stft(window="ones", center, zero pad) -> basis @ D -> halve the signal (scale=True) and the hop
      __trim_stack -> / sqrt(lengths).
    (`__early_downsample` is a no-op at sr=22050, n_bins=84: log2(nyquist / filter_cutoff) = 1.42.)
    The one step that cannot be restated is the halving resampler: librosa's default res_type "soxr_hq" is the soxr
    library.  A Kaiser-windowed half-band FIR with soxr-HQ's documented band edges stands in (same design as the
    product's, written separately here).
  * torchaudio.functional.resample (utilityFunctions.py:116-117) with its defaults (sinc_interp_hann, width 6,
    rolloff 0.99): the published kernel formula, applied the way torchaudio applies it (pad, strided correlation,
    interleave, trim to ceil(new * n / orig)).  torchaudio is absent too: unpinned.
"""
import math

import numpy as np


def halfband(att=120.0):
    """Low-pass for the 2:1 resample: pass band to 0.913 of the NEW Nyquist, stop band from the new Nyquist."""
    edge_pass, edge_stop = 0.913 / 4, 1.0 / 4                  # cycles per input sample
    width = edge_stop - edge_pass
    ntaps = int(math.ceil((att - 7.95) / (14.36 * width)))
    ntaps += 1 - ntaps % 2                                     # odd: integer group delay
    centre = (ntaps - 1) // 2
    cutoff = (edge_pass + edge_stop) / 2
    h = np.empty(ntaps)
    win = np.kaiser(ntaps, 0.1102 * (att - 8.7))
    for i in range(ntaps):
        x = i - centre
        h[i] = (2 * cutoff if x == 0 else math.sin(2 * math.pi * cutoff * x) / (math.pi * x)) * win[i]
    return h / h.sum()


def resample_half(y):
    """librosa.resample(y, orig_sr=2, target_sr=1, scale=True): ceil(n/2) samples, divided by sqrt(ratio = 1/2)."""
    h = halfband()
    c = (len(h) - 1) // 2
    full = np.convolve(y, h)                                   # full[j] = sum_k h[k] y[j-k]; symmetric h -> zero-phase at +c
    n_out = int(math.ceil(len(y) * 0.5))
    return full[c:c + 2 * n_out:2][:n_out] / math.sqrt(0.5)


def wavelet_lengths(freqs, sr, bins_per_octave, filter_scale=1.0):
    r = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r - 1) / (r + 1)                                  # filters._relative_bandwidth, equal-tempered grid
    Q = filter_scale / alpha
    cutoff = np.max(freqs * (1 + 0.5 * 1.50018310546875 / Q))  # window_bandwidth("hann")
    return Q * sr / freqs, cutoff


def octave_basis(freqs_oct, sr_oct, bins_per_octave, sparsity=0.01):
    """__vqt_filter_fft: one-sided FFT of the octave's wavelets, small entries dropped.  Returns (basis, nfft)."""
    lengths, _ = wavelet_lengths(freqs_oct, sr_oct, bins_per_octave)
    nfft = int(2.0 ** np.ceil(np.log2(lengths.max())))
    rows = []
    for ilen, f in zip(lengths, freqs_oct):
        t = np.arange(-ilen // 2, ilen // 2, dtype=float)
        sig = np.cos(2 * np.pi * f * t / sr_oct) + 1j * np.sin(2 * np.pi * f * t / sr_oct)
        n = len(sig)
        sig = sig * (0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n)))
        sig = sig / np.sum(np.abs(sig))
        left = (nfft - n) // 2
        rows.append(np.pad(sig, (left, nfft - n - left)))
    basis = np.asarray(rows) * (lengths[:, None] / float(nfft))
    fb = np.fft.fft(basis, n=nfft, axis=1)[:, :nfft // 2 + 1]
    out = np.zeros_like(fb)
    for i in range(fb.shape[0]):                               # util.sparsify_rows
        mag = np.abs(fb[i])
        order = np.sort(mag)
        cum = np.cumsum(order / mag.sum())
        thr = order[np.argmin(cum < sparsity)]
        keep = mag >= thr
        out[i, keep] = fb[i, keep]
    return out, nfft


def stft_ones(y, nfft, hop):
    """librosa.stft(y, n_fft, hop, window="ones", center=True, pad_mode="constant") -> (nfft/2+1, frames)."""
    ypad = np.pad(y, (nfft // 2, nfft // 2))
    frames = 1 + (len(ypad) - nfft) // hop
    D = np.empty((nfft // 2 + 1, frames), dtype=complex)
    for t in range(frames):
        D[:, t] = np.fft.rfft(ypad[t * hop:t * hop + nfft])
    return D


def cqt(y, sr=22050, n_bins=84, hop_length=256, bins_per_octave=12, fmin=32.70319566257483):
    """-> complex (n_bins, frames), librosa.cqt's return value."""
    y = np.asarray(y, dtype=np.float64)
    n_oct = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    _, cutoff = wavelet_lengths(freqs, sr, bins_per_octave)
    assert cutoff <= sr / 2 and hop_length % 2 ** (n_oct - 1) == 0
    resp = []
    my_y, my_sr, my_hop = y, float(sr), hop_length
    for i in range(n_oct):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        fb, nfft = octave_basis(freqs[sl], my_sr, bins_per_octave)
        fb = fb * np.sqrt(sr / my_sr)
        resp.append(fb @ stft_ones(my_y, nfft, my_hop))
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            my_y = resample_half(my_y)
    max_col = min(c.shape[-1] for c in resp)                   # __trim_stack
    out = np.empty((n_bins, max_col), dtype=complex)
    end = n_bins
    for c in resp:
        n_o = c.shape[0]
        if end < n_o:
            out[:end] = c[-end:, :max_col]
        else:
            out[end - n_o:end] = c[:, :max_col]
        end -= n_o
    lengths, _ = wavelet_lengths(freqs, sr, bins_per_octave)
    return out / np.sqrt(lengths)[:, None]


def get_cqt(waveform, sample_rate=22050, n_bins=84, hop_length=256):
    """utilityFunctions.py:39-60: -> (2, frames, n_bins) float32, [real, imaginary]."""
    c = cqt(np.squeeze(np.asarray(waveform)), sample_rate, n_bins, hop_length)
    return np.transpose(np.stack([c.real, c.imag], axis=-1), (2, 1, 0)).astype(np.float32)


def sinc_resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample defaults on a 1-D signal."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if orig == new:
        return np.asarray(x, dtype=np.float64)
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    xp = np.pad(x, (width, width + orig))
    klen = 2 * width + orig
    steps = (len(xp) - klen) // orig + 1
    out = np.empty((steps, new))
    for p in range(new):
        t = (-p / new + np.arange(-width, width + orig) / orig) * base
        t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
        win = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        tt = t * math.pi
        k = np.where(tt == 0, 1.0, np.sin(tt) / np.where(tt == 0, 1.0, tt)) * win * (base / orig)
        for i in range(steps):
            out[i, p] = np.dot(k, xp[i * orig:i * orig + klen])
    return out.reshape(-1)[:int(math.ceil(new * n / orig))]

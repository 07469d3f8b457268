"""Drop-in for the reference's dataloader.py: normalize, concat_stft_cqt, DualInstrumentDataset, custom_collate_fn,
get_dataloader, with the reference's signatures, warnings and batch layout.

The per-item path of the reference (dataloader.py:94-121) is decode -> STFT -> CQT -> normalize -> concat -> windows on
the CPU.  On this path STFT + normalize + windowing + collate layout are ONE kernel over a batch of resident waveforms
(`utilityFunctions.stft_sections`, used by `train.Trainer.set_frontend`); the functions here keep the item-wise API for
callers that want it.  File decoding + resampling (`load_audio`: torchaudio) and the CQT (`get_CQT`: librosa) are the
device restatements of cqt.py (parity unpinned -- neither library exists in any image); `DualInstrumentDataset` takes
optional `load_audio=` / `get_cqt=` callables for callers that have the real libraries."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import utilityFunctions as U
from ._lib import check, lib, ptr, stream


def normalize(x, mean, std, eps=1e-8):
    """dataloader.py:9-13: (x - mean[:, None, :]) / (std[:, None, :] + eps) for a (C,T,F) spectrogram and (C,F) statistics."""
    if x.dim() != 3 or mean.dim() != 2:
        raise ValueError(f"normalize expects x (C,T,F) and statistics (C,F), got {tuple(x.shape)} / {tuple(mean.shape)}")
    x = x.float().contiguous()
    C, T, F = x.shape
    mean, std = mean.to(x.device).float().contiguous(), std.to(x.device).float().contiguous()
    out = torch.empty_like(x)
    check(lib().ast_zscore(ptr(x), ptr(mean), ptr(std), ptr(out), C, T, F, float(eps), stream()), "ast_zscore")
    return out


def concat_stft_cqt(stft, cqt):
    """dataloader.py:15-18 (the unvalidated twin of utilityFunctions.concat_stft_cqt)."""
    return torch.cat((stft, cqt.to(stft.device)), dim=2)


class DualInstrumentDataset(Dataset):
    """dataloader.py:20-121."""

    def __init__(self, piano_dir, violin_dir, stats_path=None, use_separate_stats=True, load_audio=None, get_cqt=None,
                 device="cuda"):
        exts = (".mp3", ".wav")
        self.piano_files = sorted(os.path.join(piano_dir, f) for f in os.listdir(piano_dir) if f.endswith(exts))
        self.violin_files = sorted(os.path.join(violin_dir, f) for f in os.listdir(violin_dir) if f.endswith(exts))
        self.length = min(len(self.piano_files), len(self.violin_files))
        self.use_separate_stats = use_separate_stats
        self._load_audio, self._get_cqt, self.device = load_audio, get_cqt, device
        if use_separate_stats:
            self._load_separate_stats()
        else:
            self._load_combined_stats(stats_path)

    def _set(self, which, st):
        for k in ("stft_mean", "stft_std", "cqt_mean", "cqt_std"):
            setattr(self, f"{k}_{which}", torch.tensor(st[k]).float())

    def _load_separate_stats(self):
        pp, vp = "train_set_stats/stats_stft_cqt_piano.npz", "train_set_stats/stats_stft_cqt_violin.npz"
        if os.path.exists(pp) and os.path.exists(vp):
            self._set("piano", np.load(pp)); self._set("violin", np.load(vp))
        else:
            print("⚠️ Warning: Separate stats files not found. Using dummy normalization.")
            print(f"  Expected: {pp}, {vp}")
            self._create_dummy_separate_stats()

    def _load_combined_stats(self, stats_path):
        if stats_path is None:
            stats_path = "train_set_stats/stats_unified_stft_cqt.npz"
        if os.path.exists(stats_path):
            st = np.load(stats_path)
            self._set("piano", st); self._set("violin", st)
        else:
            print(f"⚠️ Warning: Combined stats file {stats_path} not found. Using dummy normalization.")
            self._create_dummy_separate_stats()

    def _create_dummy_separate_stats(self):
        for which in ("piano", "violin"):
            self._set(which, {"stft_mean": np.zeros((2, 513)), "stft_std": np.ones((2, 513)),
                              "cqt_mean": np.zeros((2, 84)), "cqt_std": np.ones((2, 84))})

    def __len__(self):
        return self.length

    def _item(self, path, which):
        audio, _ = (self._load_audio or U.load_audio)(path)
        audio = audio.to(self.device)
        stft = normalize(U.get_STFT(audio), getattr(self, f"stft_mean_{which}"), getattr(self, f"stft_std_{which}"))
        cqt = normalize((self._get_cqt or U.get_CQT)(audio).to(self.device), getattr(self, f"cqt_mean_{which}"), getattr(self, f"cqt_std_{which}"))
        return U.get_overlap_windows(concat_stft_cqt(stft, cqt))

    def __getitem__(self, idx):
        return {"piano": self._item(self.piano_files[idx], "piano"), "violin": self._item(self.violin_files[idx], "violin"),
                "piano_label": 0, "violin_label": 1}


def custom_collate_fn(batch):
    """dataloader.py:123-147: only the first half of the items is used, both instruments from the same items."""
    batch_size = len(batch)
    half = batch_size // 2
    piano = [batch[i]["piano"] for i in range(half)]
    violin = [batch[i]["violin"] for i in range(half)]
    out = torch.stack(piano + violin, dim=0)
    if batch_size != 2 * half:                    # odd batch: the reference leaves the last row uninitialised
        out = torch.cat([out, out.new_zeros((1,) + tuple(out.shape[1:]))], dim=0)
    labels = torch.cat([torch.zeros(half, dtype=torch.long), torch.ones(half, dtype=torch.long)])
    return out, labels


def get_dataloader(piano_dir, violin_dir, batch_size=8, shuffle=True, stats_path=None, use_separate_stats=True, **dataset_kw):
    """dataloader.py:149-172."""
    if batch_size % 2 != 0:
        print(f"Warning: batch_size={batch_size} is odd. Rounding down to {batch_size-1} for balanced batches.")
        batch_size = batch_size - 1
    dataset = DualInstrumentDataset(piano_dir, violin_dir, stats_path, use_separate_stats, **dataset_kw)
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=custom_collate_fn, drop_last=True)


# ---- variable-length clips (BASELINE configs[4]) -------------------------------------------------------------------
def sections_for_samples(n_samples: int) -> int:
    """Number of 287-frame sections get_overlap_windows makes of a clip of n_samples at hop 256 (utilityFunctions.py:246-262:
    2-3 s -> 1, 4-6 s -> 2, 7-8 s -> 3, 10 s -> 4; a tail shorter than half a window is dropped)."""
    return len(U.section_starts(1 + n_samples // 256))


class LengthBucketSampler(torch.utils.data.Sampler):
    """Batch sampler for clips of mixed length.  custom_collate_fn stacks the items of a batch (dataloader.py:136-142), so all
    of them must have the same number of sections S -- and the batched front end wants equal sample counts -- hence batches
    are drawn from one LENGTH bucket at a time (every bucket has one S; zero-padding a short clip to a longer bucket would
    move the STFT's reflect padding away from the clip's true end and change its last frames).  Yields lists of item indices;
    incomplete batches are dropped, as get_dataloader does."""

    def __init__(self, n_samples_per_item, batch_size: int, shuffle: bool = True, seed: int = 0):
        self.batch_size, self.shuffle, self.seed, self.epoch = int(batch_size), shuffle, seed, 0
        self.buckets = {}
        for i, n in enumerate(n_samples_per_item):
            self.buckets.setdefault(int(n), []).append(i)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        batches = []
        for n in sorted(self.buckets):
            idx = self.buckets[n]
            if self.shuffle:
                idx = [idx[j] for j in torch.randperm(len(idx), generator=g).tolist()]
            batches += [idx[k:k + self.batch_size] for k in range(0, len(idx) - self.batch_size + 1, self.batch_size)]
        if self.shuffle:
            batches = [batches[j] for j in torch.randperm(len(batches), generator=g).tolist()]
        return iter(batches)

    def __len__(self):
        return sum(len(v) // self.batch_size for v in self.buckets.values())

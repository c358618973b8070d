"""Batch wire format of the training loop (reference data/bedrooms.py:209-238).

The reference's DataLoader yields the tuple  (words [B,T] int64, lengths [B] int64, class_ids [B] int64,
img64, img128, img256 float32 in [-1,1])  -- train.py:111.  Everything upstream of that tuple (image folders, UMAP /
agglomerative pseudo-captions, vocabulary building) is host-side data preparation outside the hot path; what is kept here
is the tuple itself, produced either synthetically (benchmarks, tests) or from pre-resized uint8 shards on disk.

Shard format (.npz, data only): `words` int64 [N,T] (0-padded), `lengths` int64 [N], `class_ids` int64 [N],
`img64` / `img128` / `img256` uint8 [N,3,R,R].  Pixels are mapped with (x - 127.5) / 127.5 like the reference's
Training.scale_255_to_1 (utilities/training.py:23-25).
"""
from __future__ import annotations

from typing import Iterator, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

Batch = Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]


def scale_255_to_1(images: Tensor) -> Tensor:
    return (images.float() - 127.5) / 127.5


def scale_1_to_255(images: Tensor) -> Tensor:
    return images * 127.5 + 127.5


def synthetic_batches(batch_size: int, seq_len: int = 10, vocab: int = 1000, n_batches: int = 1, seed: int = 0,
                      device="cpu") -> Iterator[Batch]:
    """Random-token captions + uniform images at 64/128/256 (BASELINE.json north_star's synthetic workload)."""
    g = torch.Generator().manual_seed(seed)
    for _ in range(n_batches):
        words = torch.randint(1, vocab, (batch_size, seq_len), generator=g)
        lengths = torch.randint(2, seq_len + 1, (batch_size,), generator=g)
        words = words * (torch.arange(seq_len).view(1, -1) < lengths.view(-1, 1))          # zero-pad past each length
        class_ids = torch.randint(0, 50, (batch_size,), generator=g)
        imgs = [torch.rand(batch_size, 3, r, r, generator=g) * 2 - 1 for r in (64, 128, 256)]
        yield tuple(t.to(device) for t in (words, lengths, class_ids, *imgs))


class ShardBatches:
    """Iterates the batch tuple over one or more .npz shards; short / too-short-caption batches are skipped exactly as the
    training loop does (train.py:112: `min(lengths) < 2 or len(words) < BATCH_SIZE`)."""

    def __init__(self, paths: Sequence[str], batch_size: int, device="cpu", shuffle: bool = True, seed: int = 0):
        self.paths, self.batch_size, self.device, self.shuffle = list(paths), batch_size, device, shuffle
        self.rng = np.random.default_rng(seed)

    def __iter__(self) -> Iterator[Batch]:
        for path in self.paths:
            with np.load(path, allow_pickle=False, mmap_mode=None) as npz:
                z = {k: npz[k] for k in ("words", "lengths", "class_ids", "img64", "img128", "img256")}   # NpzFile re-reads per access
            n = len(z["lengths"])
            order = self.rng.permutation(n) if self.shuffle else np.arange(n)
            for s in range(0, n - self.batch_size + 1, self.batch_size):
                idx = np.sort(order[s:s + self.batch_size])
                lengths = torch.from_numpy(z["lengths"][idx])
                if int(lengths.min()) < 2:
                    continue
                words = torch.from_numpy(z["words"][idx])[:, :int(lengths.max())]
                class_ids = torch.from_numpy(z["class_ids"][idx])
                imgs = [scale_255_to_1(torch.from_numpy(z[f"img{r}"][idx])) for r in (64, 128, 256)]
                yield tuple(t.to(self.device, non_blocking=True) for t in (words, lengths, class_ids, *imgs))


def write_shard(path: str, words: np.ndarray, lengths: np.ndarray, class_ids: np.ndarray, img64: np.ndarray, img128: np.ndarray,
                img256: np.ndarray) -> None:
    for a, r in ((img64, 64), (img128, 128), (img256, 256)):
        if a.dtype != np.uint8 or a.shape[1:] != (3, r, r):
            raise ValueError(f"img{r} must be uint8 [N,3,{r},{r}]")
    np.savez(path, words=words.astype(np.int64), lengths=lengths.astype(np.int64), class_ids=class_ids.astype(np.int64),
             img64=img64, img128=img128, img256=img256)

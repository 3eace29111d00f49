"""Synthetic batches that honour the output contract of the reference's DataGenerator.py.

The reference's generator reads raw CSV/JSON that is not shipped (``**/data/`` is git-ignored), label-encodes
every column and adds a cumulative offset so that all features share ONE global id space
(2.FM/DataGenerator.py:76-88): field f owns the contiguous range [offset_f, offset_f + dim_f), and
``data_info.json = [dims, offsets, total]`` (:126-134).  Each feature is delivered as its own int64 ``[B,1]``
tensor keyed by name, the label as float32 ``[B,1]`` (2.FM/ModelManager.py:127-133); behaviour series are
right-padded with ``padding_index`` (5.DIN/ModelManager.py:147-149).  This module reproduces that contract
with seeded numpy generators (PCG64); ids are generated on the host and uploaded once.
"""
import numpy as np


def field_layout(total_vocab, n_fields):
    """dims (V split evenly, last field takes the remainder), offsets = (0, cumsum(dims[:-1])), total."""
    dims = [total_vocab // n_fields] * n_fields
    dims[-1] += total_vocab - sum(dims)
    offsets = [0]
    for d in dims[:-1]:
        offsets.append(offsets[-1] + d)
    return dims, offsets, total_vocab


def data_info(total_vocab, n_fields):
    """The ``data_info.json`` payload the reference's ModelManager reads (feature_info[-1] = total vocab)."""
    dims, offsets, total = field_layout(total_vocab, n_fields)
    return [dims, offsets, total]


class _Zipf:
    """Exact Zipf(s) on the finite support {0..dim-1} (rank r has weight (r+1)^-s), by inverse CDF."""

    def __init__(self, dim, s):
        w = np.arange(1, dim + 1, dtype=np.float64) ** (-s)
        self.cdf = np.cumsum(w)
        self.cdf /= self.cdf[-1]

    def draw(self, rng, size):
        return np.minimum(np.searchsorted(self.cdf, rng.random(size)), len(self.cdf) - 1).astype(np.int64)


def _draw_ids(rng, dim, size, dist, s, cache):
    if dist == "uniform":
        return rng.integers(0, dim, size=size, dtype=np.int64)
    if dist == "zipf":
        key = (dim, s)
        if key not in cache:
            cache[key] = _Zipf(dim, s)
        return cache[key].draw(rng, size)
    raise ValueError("dist must be 'uniform' or 'zipf'")


class SyntheticGenerator:
    """Batches for categorical (+ optional continuous and series) features over one global id space."""

    def __init__(self, categorical, total_vocab, continuous=(), series=(), seq_len=0, dist="uniform", zipf_s=1.05,
                 label_rate=0.25, padding_index=0, seed=0):
        self.categorical = list(categorical)
        self.continuous = list(continuous)
        self.series = list(series)
        self.seq_len = seq_len
        self.dist, self.zipf_s, self.label_rate, self.padding_index = dist, zipf_s, label_rate, padding_index
        self.total_vocab = total_vocab
        n_fields = len(self.categorical) + len(self.series)
        self.dims, self.offsets, _ = field_layout(total_vocab, n_fields)
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self._cache = {}

    def batch(self, B):
        out = {}
        f = 0
        for name in self.categorical:
            x = _draw_ids(self.rng, self.dims[f], B, self.dist, self.zipf_s, self._cache) + self.offsets[f]
            out[name] = x.reshape(B, 1)
            f += 1
        if self.series:
            lens = self.rng.integers(1, self.seq_len + 1, size=B)
            valid = np.arange(self.seq_len)[None, :] < lens[:, None]
            for name in self.series:
                # a real id never equals padding_index: shift by one inside the field when they collide
                x = _draw_ids(self.rng, self.dims[f], (B, self.seq_len), self.dist, self.zipf_s, self._cache)
                x = x + self.offsets[f]
                x = np.where(x == self.padding_index, x + 1, x)
                out[name] = np.where(valid, x, self.padding_index).astype(np.int64)
                f += 1
        for name in self.continuous:
            out[name] = self.rng.standard_normal((B, 1)).astype(np.float32)
        out["label"] = (self.rng.random((B, 1)) < self.label_rate).astype(np.float32)
        return out


def to_device(batch, device="cuda"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in batch.items()}

"""On-disk dataset format and minibatch feed of the reference, host side only.

reference: sqair/data/data.py:189-245 (pickle layout, ``load_data``, ``tensors_from_data``),
sqair/data/mnist_tools.py:37-107 (batch axes, truncation, coords padding, nums tiling, sequence-length curriculum),
sqair/index.py:224-240 (``dynamic_truncate``).

A dataset is ONE pickled dict:
    imgs    uint8   [T, N, H, W]          frames (fed as float32 / 255)
    nums    uint8   [1 or T, N, n_max+1]  prefix-ones presence (tiled over T when stored once)
    coords  float   [T, N, n_obj, 4]      (y, x, h, w) in pixels; padded with zero boxes to n_max+1 objects
    labels  int     [N, ...]              digit labels (carried, unused by the model)
The batch axis of imgs / nums / coords is 1, of labels 0.  ``sqair_amd.data.make_sequences`` produces the same dict
from procedural glyphs (MNIST itself is not available offline); files written by the reference's
``create_seq_mnist.py`` load unchanged (Python-2 pickles: ``encoding='latin1'``).
"""
from __future__ import annotations

import itertools
import pickle

import numpy as np

AXES = {"imgs": 1, "labels": 0, "nums": 1, "coords": 1}  # mnist_tools.py:37


def save_dataset(path, data):
    """Writes the dict in the reference's layout (imgs as uint8; protocol 2 so that the Python-2 reference reads it)."""
    out = dict(data)
    imgs = np.asarray(out["imgs"])
    if imgs.dtype != np.uint8:
        imgs = np.clip(np.round(imgs * 255.0), 0, 255).astype(np.uint8)
    out["imgs"] = imgs
    with open(path, "wb") as f:
        pickle.dump(out, f, protocol=2)


def load_dataset(path):
    """data.py:189-201: imgs -> float32 / 255, nums -> float32."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    data = dict(data)
    data["imgs"] = np.asarray(data["imgs"]).astype(np.float32) / 255.0
    data["nums"] = np.asarray(data["nums"]).astype(np.float32)
    return data


def process_data(data, n_timesteps=None):
    """mnist_tools.py:40-58: optional truncation to the first ``n_timesteps`` frames; coords padded with zero boxes up
    to the width of ``nums`` (n_max + 1 objects)."""
    if n_timesteps is not None:
        for k in ("imgs", "coords", "nums"):
            if k in data:
                data[k] = data[k][:n_timesteps]
    if "coords" in data:
        to_pad = data["nums"].shape[-1] - data["coords"].shape[-2]
        if to_pad > 0:
            shape = list(data["coords"].shape)
            shape[-2] = to_pad
            data["coords"] = np.concatenate((data["coords"], np.zeros(shape, dtype=data["coords"].dtype)), -2)
    return data


class MinibatchFeed(object):
    """``tensors_from_data`` (data.py:203-245) without the tf.py_func: ``shuffle=True`` draws ``batch_size`` sequence
    indices WITH replacement per batch (np.random.choice), ``shuffle=False`` cycles through consecutive full batches
    (the ragged tail is never visited).  ``next(step)`` also applies the reference's post-processing: ``nums`` stored
    once is tiled over time (mnist_tools.py:76-78) and every [T, ...] entry is truncated to the curriculum length
    ``min(seq_len + step // stage_itr, T)`` when both flags are set (mnist_tools.py:80-92)."""

    def __init__(self, data, batch_size, shuffle, seed=None, seq_len=0, stage_itr=0):
        self.data = {k: np.asarray(v) for k, v in data.items() if k in AXES}
        self.batch_size = int(batch_size)
        self.n_entries = int(self.data["imgs"].shape[AXES["imgs"]])
        if self.n_entries < self.batch_size:
            raise ValueError("dataset of {} sequences is smaller than one batch of {}".format(self.n_entries, batch_size))
        self.shuffle = bool(shuffle)
        self.rng = np.random.default_rng(seed)
        self._rolling = itertools.cycle(range(0, self.n_entries - self.batch_size + 1, self.batch_size))
        self.seq_len, self.stage_itr = int(seq_len), int(stage_itr)

    def indices(self):
        if self.shuffle:
            return self.rng.choice(self.n_entries, self.batch_size)
        start = next(self._rolling)
        return np.arange(start, start + self.batch_size)

    def next(self, step=0):
        idx = self.indices()
        batch = {k: v.take(idx, AXES[k]) for k, v in self.data.items()}
        T = batch["imgs"].shape[0]
        if batch["nums"].shape[0] != T:
            batch["nums"] = np.tile(batch["nums"], (T, 1, 1))
        if self.seq_len != 0 and self.stage_itr > 0:
            t_max = min(self.seq_len + int(step) // self.stage_itr, T)
            for k in ("imgs", "nums", "coords"):
                if k in batch:
                    batch[k] = batch[k][:t_max]
        return batch

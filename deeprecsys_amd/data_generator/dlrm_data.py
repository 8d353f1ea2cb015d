"""Synthetic inputs: the reference's "random" generator restated
(data_generator/dlrm_data_caffe2.py:34-60 dispatch, :69-124 inputs, :128-148 targets).

The engine's inputs are a pure function of numpy's legacy global RNG stream, and
request packets only carry (batch_id, batch_size), so reproducing the *stream
consumption order* exactly is part of the drop-in contract:
  per batch: rand(n, m_den) -> float32; then for every table, for every sample:
  random(L) redrawn until round(r*(size-1)) has L distinct values (np.unique: sorted).
tests/test_host_parity.py pins this against arrays captured from the reference.

`generate_fast_input_data` is NOT stream-compatible: same distribution, vectorised,
own Generator -- for benchmark-sized runs where 260k python-level draws per batch
set would dominate start-up.
"""
import sys

import numpy as np
from numpy import random as ra


class DataGenerator(object):
    def __init__(self, args):
        self.args = args

    def generate_input_data(self):
        raise NotImplementedError

    def generate_output_data(self):
        raise NotImplementedError


class DLRMDataGenerator(DataGenerator):
    def generate_input_data(self):
        a = self.args
        ln_bot = np.array(a.arch_mlp_bot.split("-"), dtype=int)
        if a.data_generation != "random":
            # the reference's "synthetic" branch is unreachable (it calls a method as a free
            # function, dlrm_data_caffe2.py:51) and "dataset" exits; keep both as hard errors
            sys.exit("ERROR: --data_generation=" + a.data_generation + " is not supported")
        ln_emb = np.array(a.arch_embedding_size.split("-"), dtype=int)
        return self.generate_random_input_data(a.num_batches, a.max_mini_batch_size, a.round_targets,
                                               a.num_indices_per_lookup,
                                               a.num_indices_per_lookup_fixed, ln_bot[0], ln_emb)

    def generate_output_data(self):
        a = self.args
        return self.generate_random_output_data(a.num_batches, a.max_mini_batch_size,
                                                round_targets=a.round_targets)

    def generate_random_input_data(self, num_batches, mini_batch_size, round_targets,
                                   num_indices_per_lookup, num_indices_per_lookup_fixed, m_den,
                                   ln_emb):
        lX, lS_lengths, lS_indices = [], [], []
        L = np.int32(num_indices_per_lookup)
        for _ in range(num_batches):
            n = mini_batch_size
            lX.append(ra.rand(n, m_den).astype(np.float32))
            emb_lengths, emb_indices = [], []
            for size in ln_emb:
                lengths, indices = [], []
                scale = size - 1
                for _s in range(n):
                    while True:  # redraw until L distinct rows (consumes L doubles per try)
                        group = np.unique(np.round(ra.random(L) * scale).astype(np.int32))
                        if group.size == L:
                            break
                    lengths.append(np.int32(group.size))
                    indices += group.tolist()
                emb_lengths.append(lengths)
                emb_indices.append(indices)
            lS_lengths.append(emb_lengths)
            lS_indices.append(emb_indices)
        return (num_batches, lX, lS_lengths, lS_indices)

    def generate_random_output_data(self, num_batches, mini_batch_size, num_targets=1,
                                    round_targets=False):
        lT = []
        for _ in range(num_batches):
            P = ra.rand(mini_batch_size, num_targets).astype(np.float32)
            if round_targets:
                P = np.round(P).astype(np.int32)
            lT.append(P)
        return (num_batches, lT)


def generate_fast_input_data(num_batches, n, m_den, ln_emb, L, seed):
    """Same distribution as the reference generator (uniform dense; per bag L distinct,
    sorted rows drawn as round(u*(size-1))), vectorised; not stream-compatible."""
    rng = np.random.default_rng(seed)
    lX, lS_l, lS_i = [], [], []
    for _ in range(num_batches):
        lX.append(rng.random((n, m_den), dtype=np.float32))
        lens, idxs = [], []
        for size in ln_emb:
            rows = np.sort(np.round(rng.random((n, L)) * (size - 1)).astype(np.int64), axis=1)
            dup = (np.diff(rows, axis=1) == 0).any(axis=1) if L > 1 else np.zeros(n, bool)
            while dup.any():
                k = int(dup.sum())
                rows[dup] = np.sort(np.round(rng.random((k, L)) * (size - 1)).astype(np.int64), axis=1)
                dup = (np.diff(rows, axis=1) == 0).any(axis=1)
            lens.append(np.full(n, L, dtype=np.int32))
            idxs.append(rows.reshape(-1))
        lS_l.append(lens)
        lS_i.append(idxs)
    return (num_batches, lX, lS_l, lS_i)

"""Synthetic batches with the reference loader's batch contract (loader/base.py:436-450,
make_loader.py:4-53): the HuggingFace datasets the reference trains on are not reachable offline.

Recipe: host generator seeded 1000*rank + step; spikes ~ Poisson(0.3) counts,
behaviour ~ N(0,1), attention masks all ones (or right-padded), time stamps arange(T).
"""
from __future__ import annotations

from typing import List, Optional

import torch


def synth_batch(B, T, n_ap, n_beh, seed, pad: Optional[List[int]] = None, eid="synthetic", full_contract=True):
    g = torch.Generator().manual_seed(seed)
    spikes = torch.poisson(torch.full((B, T, n_ap), 0.3), generator=g)
    beh = torch.randn(B, T, n_beh, generator=g)
    attn = torch.ones(B, T, dtype=torch.int64)
    if pad is not None:
        for b, nb in enumerate(pad):
            if nb:
                attn[b, T - nb:] = 0
    batch = dict(spikes_data=spikes, target=beh, time_attn_mask=attn,
                 spikes_timestamps=torch.arange(T, dtype=torch.int64)[None].repeat(B, 1))
    if full_contract:
        batch.update(space_attn_mask=torch.ones(B, n_ap, dtype=torch.int64),
                     spikes_spacestamps=torch.arange(n_ap, dtype=torch.int64)[None].repeat(B, 1),
                     neuron_depths=torch.zeros(B, n_ap), neuron_regions=[["XX"] * B for _ in range(n_ap)],
                     eid=[eid] * B, choice=torch.zeros(B), block=torch.zeros(B), reward=torch.zeros(B))
    return batch


class SyntheticLoader:
    """Iterable of `n_batches` synthetic batches; deterministic per (rank, step)."""

    def __init__(self, n_batches, batch_size, T=100, n_ap=668, n_beh=2, rank=0, seed0=0, device=None, cache=False):
        self.n, self.B, self.T, self.n_ap, self.n_beh = n_batches, batch_size, T, n_ap, n_beh
        self.rank, self.seed0, self.device = rank, seed0, device
        self._cache = {} if cache else None

    def __len__(self):
        return self.n

    def _make(self, step):
        b = synth_batch(self.B, self.T, self.n_ap, self.n_beh, seed=1000 * self.rank + self.seed0 + step)
        if self.device is not None:
            b = {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        return b

    def __iter__(self):
        for step in range(self.n):
            if self._cache is None:
                yield self._make(step)
            else:
                if step not in self._cache:
                    self._cache[step] = self._make(step)
                yield dict(self._cache[step])

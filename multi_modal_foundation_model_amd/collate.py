"""GPU collate for IBL-style trials: what `make_loader(...)`'s dataset + default collate produce
(reference loader/base.py:304-450, loader/make_loader.py:4-53), with the densify / pad / mask work done by
`mmfm_collate_csr` on the device instead of per-trial numpy on the host.

    batch = collate_ibl_trials(trials, target=["wheel-speed", "whisker-motion-energy"], max_time_length=100,
                               max_space_length=668, pad_value=-1., device="cuda")

`trials` are dicts with the HuggingFace dataset columns (`spikes_sparse_data/indices/indptr/shape`, behaviour
columns, `cluster_depths`, `cluster_regions`, `eid`, `choice`, `block`, `reward`).  Only the `pad_to_right=True`,
unsorted, un-stitched path of the reference is covered (the one `train_multi_modal.py` uses).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops as K


def collate_ibl_trials(trials, target, max_time_length, max_space_length, pad_value=0.0, device="cuda", load_meta=True):
    B, max_T, max_N = len(trials), int(max_time_length), int(max_space_length)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("collate_ibl_trials runs mmfm_collate_csr on the GPU; there is no CPU path in this package")
    def counts(t):
        """The IBL dataset stores the CSR values as ubyte; anything that would not survive the cast is refused, not wrapped."""
        a = np.asarray(t["spikes_sparse_data"])
        if a.size and (not np.issubdtype(a.dtype, np.integer) and np.any(a != np.floor(a)) or a.min() < 0 or a.max() > 255):
            raise ValueError("collate_ibl_trials: spikes_sparse_data must hold integer counts in [0, 255] (uint8 CSR values, "
                             f"as in the IBL datasets); got dtype {a.dtype}, range [{a.min()}, {a.max()}]")
        return a.astype(np.uint8)
    data = np.concatenate([counts(t) for t in trials]) if B else np.zeros(0, np.uint8)
    idx = np.concatenate([np.asarray(t["spikes_sparse_indices"], dtype=np.int32) for t in trials])
    ptr = np.concatenate([np.asarray(t["spikes_sparse_indptr"], dtype=np.int64) for t in trials])
    T_b = np.asarray([t["spikes_sparse_shape"][0] for t in trials], dtype=np.int32)
    N_b = np.asarray([t["spikes_sparse_shape"][1] for t in trials], dtype=np.int32)
    nnz = np.asarray([len(t["spikes_sparse_data"]) for t in trials], dtype=np.int64)
    nnz_off = np.concatenate([[0], np.cumsum(nnz)[:-1]]).astype(np.int64)
    ptr_off = np.concatenate([[0], np.cumsum(T_b.astype(np.int64) + 1)[:-1]]).astype(np.int64)
    if len(data) == 0:                                  # keep valid device pointers for the all-empty batch
        data, idx = np.zeros(1, np.uint8), np.zeros(1, np.int32)
    to = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
    d_data, d_idx, d_ptr, d_po, d_no, d_T, d_N = map(to, (data, idx, ptr, ptr_off, nnz_off, T_b, N_b))
    spikes = torch.empty(B, max_T, max_N, device=dev)
    tmask = torch.empty(B, max_T, dtype=torch.int64, device=dev)
    smask = torch.empty(B, max_N, dtype=torch.int64, device=dev)
    K.collate_csr(B, max_T, max_N, pad_value, d_data, d_idx, d_ptr, d_po, d_no, d_T, d_N, spikes, tmask, smask)
    batch = dict(spikes_data=spikes, time_attn_mask=tmask, space_attn_mask=smask,
                 spikes_timestamps=torch.arange(max_T, device=dev)[None].repeat(B, 1),
                 spikes_spacestamps=torch.arange(max_N, device=dev)[None].repeat(B, 1),
                 eid=[t["eid"] for t in trials])
    if target is not None:      # behaviours are NOT padded by the reference (loader/base.py:320-327): equal lengths required to stack
        tg = np.stack([np.stack([np.asarray(t[b], dtype=np.float32) for b in target]).T for t in trials])
        batch["target"] = to(tg)
    for k in ("choice", "block", "reward"):
        batch[k] = to(np.asarray([t[k] for t in trials], dtype=np.float32))
    if load_meta:
        depths = np.full((B, max_N), np.nan, dtype=np.float32)
        regions = [["nan"] * B for _ in range(max_N)]            # default_collate layout: list of N lists of B strings
        for b, t in enumerate(trials):
            n = min(int(N_b[b]), max_N)
            depths[b, :n] = np.asarray(t["cluster_depths"], dtype=np.float32)[:n]
            for j in range(n):
                regions[j][b] = str(t["cluster_regions"][j])
        batch["neuron_depths"], batch["neuron_regions"] = torch.from_numpy(depths), regions
    return batch

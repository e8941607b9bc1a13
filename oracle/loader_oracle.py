"""CPU oracle for the loader's per-trial preprocessing.

TEST INFRASTRUCTURE ONLY (see oracle/mm_oracle.py).  Parity status: PINNED against
tests/golden/loader_collate.npz, produced by the reference's own `BaseDataset.__getitem__`.

Restates, in numpy, what `BaseDataset._preprocess_ibl_data` does on the path the entry script uses
(`pad_to_right=True`, no sorting, no stitching, no nemo; loader/base.py:304-450) and
`get_binned_spikes_from_sparse` (utils/dataset_utils.py:38-43).
"""
import numpy as np


def csr_to_dense(data, indices, indptr, shape):
    """scipy.sparse.csr_array(...).toarray(): duplicate (row, col) entries ADD (dataset_utils.py:38-43)."""
    T, N = int(shape[0]), int(shape[1])
    out = np.zeros((T, N), dtype=np.asarray(data).dtype if len(data) else np.uint8)
    for r in range(T):
        for j in range(int(indptr[r]), int(indptr[r + 1])):
            out[r, int(indices[j])] += data[j]
    return out


def attention_mask(seq_len, pad_len):
    """loader/base.py:77-86."""
    m = np.ones(seq_len)
    if pad_len:
        m[-pad_len:] = 0
    return m


def preprocess_trial(trial, target, max_T, max_N, pad_value):
    """loader/base.py:304-450 for one trial dict -> the per-sample dict of the batch contract."""
    x = csr_to_dense(trial["spikes_sparse_data"], trial["spikes_sparse_indices"], trial["spikes_sparse_indptr"],
                     trial["spikes_sparse_shape"]).astype(np.float64)
    T_i, N_i = x.shape
    depths = np.asarray(trial["cluster_depths"], dtype=np.float32)
    regions = [str(r) for r in trial["cluster_regions"]]
    pad_t = pad_n = 0
    if T_i > max_T:                                             # :389-397
        x = x[:max_T]
    else:
        pad_t = max_T - T_i
        x = np.concatenate([x, np.ones((pad_t, N_i)) * pad_value], 0) if pad_t else x
    if N_i > max_N:                                             # :399-425
        x, depths, regions = x[:, :max_N], depths[:max_N], regions[:max_N]
    else:
        pad_n = max_N - N_i
        if pad_n:
            x = np.concatenate([x, np.ones((x.shape[0], pad_n)) * pad_value], 1)
            depths = np.concatenate([depths, np.ones(pad_n) * np.nan])
            regions = regions + ["nan"] * pad_n
    tgt = np.array([np.asarray(trial[b], dtype=np.float32) for b in target]).T if target else np.array([np.nan])
    return dict(spikes_data=x.astype(np.float32), time_attn_mask=attention_mask(max_T, pad_t).astype(np.int64),
                space_attn_mask=attention_mask(max_N, pad_n).astype(np.int64), spikes_timestamps=np.arange(max_T).astype(np.int64),
                spikes_spacestamps=np.arange(max_N).astype(np.int64), target=tgt, neuron_depths=depths, neuron_regions=regions,
                eid=trial["eid"], choice=np.float32(trial["choice"]), block=np.float32(trial["block"]), reward=np.float32(trial["reward"]))


def synth_session_trials(n_neurons, n_trials, T, seed, eid):
    """Synthetic IBL-style trials of ONE session (HuggingFace column layout, loader/base.py:304-327): Poisson-like
    sparse counts as CSR uint8, two behaviour traces of length T.  Shared by oracle/make_goldens.py and the tests so
    the reference and the build see identical trials (BASELINE configs[2]: multi-session, neurons right-padded)."""
    rng = np.random.default_rng(seed)
    trials = []
    for i in range(n_trials):
        dense = ((rng.random((T, n_neurons)) < 0.25) * rng.integers(1, 4, (T, n_neurons))).astype(np.uint8)
        indptr, indices, data = [0], [], []
        for r in range(T):
            nz = np.nonzero(dense[r])[0]
            indices += nz.tolist()
            data += dense[r, nz].tolist()
            indptr.append(len(indices))
        d = dict(spikes_sparse_data=data, spikes_sparse_indices=indices, spikes_sparse_indptr=indptr,
                 spikes_sparse_shape=[T, n_neurons], choice=float(i % 2), block=0.2, reward=1.0, eid=eid,
                 cluster_depths=rng.random(n_neurons).tolist(), cluster_regions=[f"R{j % 3}" for j in range(n_neurons)])
        d["wheel-speed"] = rng.standard_normal(T).astype(np.float32).tolist()
        d["whisker-motion-energy"] = rng.standard_normal(T).astype(np.float32).tolist()
        trials.append(d)
    return trials


def collate(trials, target, max_T, max_N, pad_value):
    """torch default_collate over preprocess_trial outputs, as numpy: stack along a new batch axis; `neuron_regions`
    becomes a list of N lists of B strings (loader/make_loader.py:51)."""
    outs = [preprocess_trial(t, target, max_T, max_N, pad_value) for t in trials]
    batch = {k: np.stack([o[k] for o in outs]) for k in ("spikes_data", "time_attn_mask", "space_attn_mask", "spikes_timestamps",
                                                          "spikes_spacestamps", "target", "neuron_depths")}
    batch["neuron_regions"] = [[o["neuron_regions"][j] for o in outs] for j in range(max_N)]
    batch["eid"] = [o["eid"] for o in outs]
    return batch

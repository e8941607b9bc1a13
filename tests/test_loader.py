"""loader collate.  CPU: the numpy oracle vs the fixture produced by the reference's own
BaseDataset; GPU: mmfm_collate_csr (through collate_ibl_trials) vs the oracle, bit-exact (integer counts)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz
from oracle import loader_oracle as LO

TARGET = ["wheel-speed", "whisker-motion-energy"]


def fixture_trials():
    z, meta = load_npz("loader_collate.npz")
    trials = []
    for i, m in enumerate(meta["trials"]):
        t = {k: z[f"t{i}/in/{k}"].tolist() for k in ("spikes_sparse_data", "spikes_sparse_indices", "spikes_sparse_indptr",
                                                      "spikes_sparse_shape", "wheel-speed", "whisker-motion-energy", "cluster_depths")}
        t.update(cluster_regions=m["regions_in"], eid=m["eid"], choice=float(z[f"t{i}/out/choice"]), block=float(z[f"t{i}/out/block"]),
                 reward=float(z[f"t{i}/out/reward"]))
        trials.append(t)
    return z, meta, trials


def test_loader_oracle_matches_reference_fixture():
    z, meta, trials = fixture_trials()
    for i, t in enumerate(trials):
        out = LO.preprocess_trial(t, TARGET, meta["max_T"], meta["max_N"], meta["pad"])
        for k in ("spikes_data", "time_attn_mask", "space_attn_mask", "spikes_timestamps", "spikes_spacestamps", "target"):
            np.testing.assert_array_equal(out[k], z[f"t{i}/out/{k}"], err_msg=f"trial {i} {k}")
            assert out[k].dtype == z[f"t{i}/out/{k}"].dtype, (i, k)
        np.testing.assert_array_equal(np.nan_to_num(out["neuron_depths"], nan=-7), np.nan_to_num(z[f"t{i}/out/neuron_depths"], nan=-7))
        assert out["neuron_regions"] == meta["trials"][i]["regions_out"]


def test_csr_duplicates_add_like_scipy():
    from scipy.sparse import csr_array
    data, idx, ptr = [1, 2, 3, 4], [0, 0, 2, 1], [0, 3, 4]
    np.testing.assert_array_equal(LO.csr_to_dense(data, idx, ptr, (2, 3)), csr_array((data, idx, ptr), shape=(2, 3)).toarray())


@pytest.mark.gpu
def test_gpu_collate_bit_exact_vs_oracle_and_fixture():
    from multi_modal_foundation_model_amd.collate import collate_ibl_trials
    z, meta, trials = fixture_trials()
    same_T = [t for t in trials if t["spikes_sparse_shape"][0] == 10]          # behaviours stack only for equal lengths
    batch = collate_ibl_trials(same_T, TARGET, meta["max_T"], meta["max_N"], meta["pad"], device="cuda")
    ids = [i for i, t in enumerate(trials) if t["spikes_sparse_shape"][0] == 10]
    for b, i in enumerate(ids):
        for k in ("spikes_data", "time_attn_mask", "space_attn_mask", "spikes_timestamps", "spikes_spacestamps", "target"):
            np.testing.assert_array_equal(batch[k][b].cpu().numpy(), z[f"t{i}/out/{k}"], err_msg=f"trial {i} {k}")
        assert [r[b] for r in batch["neuron_regions"]] == meta["trials"][i]["regions_out"]
    # ragged lengths (spikes only), duplicates, an empty trial, truncation in both dimensions, larger shapes
    rng = np.random.default_rng(0)
    big = []
    for (T_i, N_i) in [(100, 668), (37, 300), (120, 700), (100, 1), (5, 668), (64, 64)]:
        nnz = int(T_i * N_i * 0.08)
        rows = np.sort(rng.integers(0, T_i, nnz))
        cols = rng.integers(0, N_i, nnz)                                       # duplicates on purpose
        ptr = np.searchsorted(rows, np.arange(T_i + 1)).tolist()
        big.append(dict(spikes_sparse_data=rng.integers(1, 5, nnz).tolist(), spikes_sparse_indices=cols.tolist(), spikes_sparse_indptr=ptr,
                        spikes_sparse_shape=[T_i, N_i], cluster_depths=rng.random(N_i).tolist(), cluster_regions=["XX"] * N_i,
                        eid="e", choice=0.0, block=0.5, reward=1.0))
    big.append(dict(spikes_sparse_data=[], spikes_sparse_indices=[], spikes_sparse_indptr=[0] * 11, spikes_sparse_shape=[10, 20],
                    cluster_depths=[0.0] * 20, cluster_regions=["XX"] * 20, eid="e", choice=0.0, block=0.5, reward=1.0))
    batch = collate_ibl_trials(big, None, 100, 668, -1.0, device="cuda")
    for b, t in enumerate(big):
        ref = LO.preprocess_trial(dict(t, **{"wheel-speed": [], "whisker-motion-energy": []}), [], 100, 668, -1.0)
        np.testing.assert_array_equal(batch["spikes_data"][b].cpu().numpy(), ref["spikes_data"], err_msg=f"big trial {b}")
        np.testing.assert_array_equal(batch["time_attn_mask"][b].cpu().numpy(), ref["time_attn_mask"])
        np.testing.assert_array_equal(batch["space_attn_mask"][b].cpu().numpy(), ref["space_attn_mask"])


# ------------------------------------------------------------------ BASELINE configs[2]: multi-session, padded neurons
def _multisession_batches():
    g = load_json("multisession_curve.json")
    sessions = [LO.synth_session_trials(n, g["trials"], g["T"], seed=100 + i, eid=f"session{i}") for i, n in enumerate(g["neurons"])]
    return g, sessions


def test_multisession_oracle_vs_reference_fixture():
    """Reference make_loader -> model over 6 sessions of 6..14 neurons padded to 14 with -1 (12 AdamW steps) against the
    oracle's loader restatement + model restatement."""
    import torch
    from oracle import mm_oracle as O
    from test_oracle_golden import default_masker_cfg
    g, sessions = _multisession_batches()
    cfg = O.OracleCfg(hidden=32, heads=4, inter=64, n_enc=1, n_dec=1, max_F=8, embed_dropout=0.0, dropout=0.0,
                      channels={"ap": g["max_N"], "behavior": 2})
    total = g["epochs"] * len(sessions)
    tr = O.OracleTrainer(O.init_state_dict(cfg, seed=g["model_seed"]), cfg, default_masker_cfg(), total_steps=total)
    torch.manual_seed(1234)
    for s in range(total):
        nb = LO.collate(sessions[s % len(sessions)], TARGET, g["T"], g["max_N"], g["pad"])
        batch = dict(spikes_data=torch.from_numpy(nb["spikes_data"]), target=torch.from_numpy(nb["target"]).float(),
                     time_attn_mask=torch.from_numpy(nb["time_attn_mask"]), spikes_timestamps=torch.from_numpy(nb["spikes_timestamps"]))
        loss = tr.step(batch, g["objective"][s])
        assert loss.item() == pytest.approx(g["loss"][s], rel=1e-4), s


@pytest.mark.gpu
def test_multisession_gpu_collate_and_model_vs_reference_fixture():
    """The same 12 steps on the MI355X: mmfm_collate_csr densifies/pads each single-session batch on the device and the
    HIP engine trains on it; loss curve against the reference's (rtol 1e-4) and final parameter norms."""
    import torch
    from helpers import build_model, make_optimizer, tiny_config
    from multi_modal_foundation_model_amd.collate import collate_ibl_trials
    from oracle import mm_oracle as O
    g, sessions = _multisession_batches()
    model = build_model(tiny_config(), g["max_N"], 2, seed=g["model_seed"]).cuda().train()
    total = g["epochs"] * len(sessions)
    opt, sch = make_optimizer(model, total)
    torch.manual_seed(1234)
    losses = []
    for s in range(total):
        batch = collate_ibl_trials(sessions[s % len(sessions)], TARGET, g["T"], g["max_N"], g["pad"], device="cuda")
        md = O.make_mod_dict({k: batch[k] for k in ("spikes_data", "target", "time_attn_mask", "spikes_timestamps")}, g["objective"][s])
        for d in md.values():
            d["targets_modality"], d["targets_timestamp"] = d["inputs_modality"], d["inputs_timestamp"]
        out = model(md)
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(out.loss.detach())
        assert {k: int(v) for k, v in out.mod_n_examples.items()} == g["n"][s]
    np.testing.assert_allclose([x.item() for x in losses], g["loss"], rtol=1e-4)
    for k, v in model.state_dict().items():
        assert float(v.double().norm()) == pytest.approx(g["final_norm"][k], rel=1e-4, abs=1e-7), k


# ------------------------------------------------------------------ BASELINE configs[2] at its stated size: 40 sessions, 300-668 neurons
def _big_sessions(g, upto=None):
    return [LO.synth_session_trials(n, g["trials"], g["T"], seed=500 + i, eid=f"session{i}") for i, n in enumerate(g["neurons"][:upto])]


def test_multisession_big_oracle_prefix_vs_reference_fixture():
    """40 sessions of 300-668 neurons right-padded with -1 to 668, default model (oracle/make_goldens.py:fx_multisession_big):
    the oracle's loader + model restatement replays the first steps of the reference's pass over the sessions."""
    import torch
    from oracle import mm_oracle as O
    from test_oracle_golden import default_masker_cfg
    g = load_json("multisession_big.json")
    assert len(g["neurons"]) == 40 and min(g["neurons"]) >= 300 and max(g["neurons"]) <= 668
    cfg = O.OracleCfg(embed_dropout=0.0, dropout=0.0)
    tr = O.OracleTrainer(O.init_state_dict(cfg, seed=g["model_seed"]), cfg, default_masker_cfg(), total_steps=g["sessions"])
    torch.manual_seed(1234)
    for s, trials in enumerate(_big_sessions(g, upto=3)):
        nb = LO.collate(trials, TARGET, g["T"], g["max_N"], g["pad"])
        batch = dict(spikes_data=torch.from_numpy(nb["spikes_data"]), target=torch.from_numpy(nb["target"]).float(),
                     time_attn_mask=torch.from_numpy(nb["time_attn_mask"]), spikes_timestamps=torch.from_numpy(nb["spikes_timestamps"]))
        assert tr.step(batch, g["objective"][s]).item() == pytest.approx(g["loss"][s], rel=1e-4), s


@pytest.mark.gpu
def test_multisession_big_gpu_vs_reference_fixture():
    """All 40 steps on the MI355X in fp32 parity mode: device-side collate of every session, loss curve (rtol 1e-4), the exact
    masked-element counts and the final parameter norms against the reference's run."""
    import torch
    from helpers import build_model, make_optimizer, model_config
    from multi_modal_foundation_model_amd.collate import collate_ibl_trials
    from oracle import mm_oracle as O
    g = load_json("multisession_big.json")
    model = build_model(model_config(dropout=0.0, emb_dropout=0.0), g["max_N"], 2, seed=g["model_seed"]).cuda().train()
    opt, sch = make_optimizer(model, g["sessions"])
    torch.manual_seed(1234)
    losses = []
    for s, trials in enumerate(_big_sessions(g)):
        batch = collate_ibl_trials(trials, TARGET, g["T"], g["max_N"], g["pad"], device="cuda")
        md = O.make_mod_dict({k: batch[k] for k in ("spikes_data", "target", "time_attn_mask", "spikes_timestamps")}, g["objective"][s])
        for d in md.values():
            d["targets_modality"], d["targets_timestamp"] = d["inputs_modality"], d["inputs_timestamp"]
        out = model(md)
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(out.loss.detach())
        assert {k: int(v) for k, v in out.mod_n_examples.items()} == g["n"][s]
    np.testing.assert_allclose([x.item() for x in losses], g["loss"], rtol=1e-4)
    for k, v in model.state_dict().items():
        assert float(v.double().norm()) == pytest.approx(g["final_norm"][k], rel=1e-4, abs=1e-7), k


def test_collate_refuses_counts_that_do_not_fit_uint8():
    """The CSR values are ubyte upstream (IBL datasets); a silent wrap of 300 -> 44 would corrupt the spikes."""
    from multi_modal_foundation_model_amd.collate import collate_ibl_trials
    t = LO.synth_session_trials(5, 1, 4, seed=1, eid="s")[0]
    bad = dict(t, spikes_sparse_data=[300] + list(t["spikes_sparse_data"])[1:])
    with pytest.raises((ValueError, RuntimeError)):
        collate_ibl_trials([bad], TARGET, 4, 5, -1.0, device="cuda" if __import__("torch").cuda.is_available() else "cpu")

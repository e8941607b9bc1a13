"""evaluation metrics.  CPU: the numpy oracle vs the fixture produced by the reference's own
bits_per_spike / neg_log_likelihood, and vs scikit-learn for R^2 (torcheval absent: see oracle/metrics_oracle.py).
GPU: mmfm_bits_per_spike / mmfm_r2_series through the C-ABI vs the oracle and the fixture."""
import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import metrics_oracle as MO


def test_bits_per_spike_oracle_vs_reference_fixture():
    z, meta = load_npz("eval_metrics.npz")
    for c in meta["cases"]:
        r, s = z[f"c{c['id']}/rates"].astype(np.float64), z[f"c{c['id']}/spikes"].astype(np.float64)
        assert MO.neg_log_likelihood(r, s) == pytest.approx(c["nll"], rel=1e-12)
        assert MO.bits_per_spike(r, s) == pytest.approx(c["bps"], rel=1e-12)


def test_bits_per_spike_per_neuron_oracle_vs_reference_fixture():
    z, meta = load_npz("eval_metrics.npz")
    for c in meta["cases"]:
        r, s = z[f"c{c['id']}/rates"].astype(np.float64), z[f"c{c['id']}/spikes"].astype(np.float64)
        np.testing.assert_allclose(MO.bits_per_spike_per_neuron(r, s), z[f"c{c['id']}/bps_per_neuron"], rtol=1e-12, equal_nan=True)


def test_heldout_mask_bit_exact_vs_reference_fixture():
    """utils/eval_utils.py:988-1045, every evaluation mode: zeroed spikes, eval mask and held-out indices, exactly."""
    import json
    from utils.eval_utils import heldout_mask
    z, meta = load_npz("eval_metrics.npz")
    sp = torch.from_numpy(z["hm/spikes"])
    regions = np.array(json.loads(bytes(z["hm/regions"]).decode()))
    for j, kw in enumerate(meta["heldout"]):
        call = {k: (np.array(v, dtype=np.int64) if k == "heldout_idxs" else v) for k, v in kw.items()}
        out = heldout_mask(sp.clone(), neuron_regions=regions, **call)
        np.testing.assert_array_equal(out["spikes"].numpy(), z[f"hm/{j}/spikes"], err_msg=str(kw))
        np.testing.assert_array_equal(out["eval_mask"].numpy(), z[f"hm/{j}/eval_mask"], err_msg=str(kw))
        assert out["eval_mask"].dtype == torch.int64
        np.testing.assert_array_equal(np.asarray(out["heldout_idxs"], dtype=np.int64), z[f"hm/{j}/heldout_idxs"], err_msg=str(kw))
    with pytest.raises(NotImplementedError):
        heldout_mask(sp, mode="per_neuron")


def test_r2_oracle_vs_sklearn():
    from sklearn.metrics import r2_score
    rng = np.random.default_rng(0)
    gt, pred = rng.standard_normal((6, 40, 5)), rng.standard_normal((6, 40, 5))
    got = MO.r2_series(gt, pred)
    for g in range(6):
        for c in range(5):
            assert got[g, c] == pytest.approx(r2_score(gt[g, :, c], pred[g, :, c]), rel=1e-12)
    gt[2, :, 1] = 3.0                                        # constant series -> invalid, masked out of the trial average
    vals = MO.r2_series(gt, pred)
    assert not np.isfinite(vals[2, 1])
    expect = np.mean([np.mean([v for v in row if np.isfinite(v)]) for row in vals])
    assert MO.trial_avg_r2(gt, pred) == pytest.approx(expect, rel=1e-12)


@pytest.mark.gpu
def test_gpu_bits_per_spike_vs_reference_fixture():
    from multi_modal_foundation_model_amd.metrics import bits_per_spike
    z, meta = load_npz("eval_metrics.npz")
    for c in meta["cases"]:
        r, s = torch.from_numpy(z[f"c{c['id']}/rates"]).cuda(), torch.from_numpy(z[f"c{c['id']}/spikes"]).cuda()
        assert bits_per_spike(r, s) == pytest.approx(c["bps"], rel=2e-5, abs=1e-6)      # fp64 sums, fp32 result
    # eval-sized problem (B = 512 trials x 100 bins x 668 neurons) against the oracle
    g = torch.Generator().manual_seed(1)
    s = torch.poisson(torch.full((512, 100, 668), 0.3), generator=g)
    r = torch.exp(torch.randn(512, 100, 668, generator=g) * 0.3 - 1.2)
    assert bits_per_spike(r.cuda(), s.cuda()) == pytest.approx(MO.bits_per_spike(r.numpy(), s.numpy()), rel=2e-5)


@pytest.mark.gpu
def test_gpu_bits_per_spike_per_neuron_vs_reference_fixture():
    from multi_modal_foundation_model_amd.metrics import bits_per_spike_per_neuron
    z, meta = load_npz("eval_metrics.npz")
    for c in meta["cases"]:
        r, s = torch.from_numpy(z[f"c{c['id']}/rates"]).cuda(), torch.from_numpy(z[f"c{c['id']}/spikes"]).cuda()
        got = bits_per_spike_per_neuron(r, s).cpu().numpy()
        np.testing.assert_allclose(got, z[f"c{c['id']}/bps_per_neuron"], rtol=3e-5, atol=1e-6, equal_nan=True)
    from utils.eval_utils import bits_per_spike_per_neuron as bpsn_np, bits_per_spike as bps_np       # numpy in, like upstream
    c = meta["cases"][1]
    np.testing.assert_allclose(bpsn_np(z["c1/rates"], z["c1/spikes"]), z["c1/bps_per_neuron"], rtol=3e-5, atol=1e-6, equal_nan=True)
    assert bps_np(z["c1/rates"], z["c1/spikes"]) == pytest.approx(c["bps"], rel=2e-5)
    g = torch.Generator().manual_seed(3)                       # eval-sized: 512 trials x 100 bins x 668 neurons
    s = torch.poisson(torch.full((512, 100, 668), 0.3), generator=g)
    r = torch.exp(torch.randn(512, 100, 668, generator=g) * 0.3 - 1.2)
    np.testing.assert_allclose(bits_per_spike_per_neuron(r.cuda(), s.cuda()).cpu().numpy(), MO.bits_per_spike_per_neuron(r.numpy(), s.numpy()),
                               rtol=3e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_r2_series_and_metrics_list():
    from multi_modal_foundation_model_amd.metrics import r2_series, trial_avg_r2
    from utils.utils import metrics_list
    g = torch.Generator().manual_seed(2)
    base_gt, base_pr = torch.randn(40, 100, 50, generator=g), torch.randn(40, 100, 50, generator=g)
    base_gt[3, :, 7] = 0.25                                  # a constant series
    # the trainer's view: [B, T, 50].transpose(-1, 0) = [50, T, B], non-contiguous (trainer/base.py:252-256)
    gt, pr = base_gt.cuda().transpose(-1, 0), base_pr.cuda().transpose(-1, 0)
    got = r2_series(gt, pr).cpu().numpy()
    ref = MO.r2_series(gt.cpu().numpy(), pr.cpu().numpy())
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all()
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-5, atol=1e-6)
    want = MO.trial_avg_r2(gt.cpu().numpy(), pr.cpu().numpy())
    assert trial_avg_r2(gt, pr) == pytest.approx(want, rel=1e-5)
    assert metrics_list(gt, pr, metrics=["r2"])["r2"] == pytest.approx(want, rel=1e-5)       # device path of the API mirror
    assert metrics_list(gt.cpu(), pr.cpu(), metrics=["r2"])["r2"] == pytest.approx(want, rel=1e-9)   # host path unchanged


@pytest.mark.gpu
def test_eval_driver_vs_reference_core_fixture():
    """the forward-only evaluation driver (held-out mask -> engine eval plan -> exp -> per-neuron bits/spike on the
    device) against the same pass assembled from the reference's own pieces (oracle/make_goldens.py:fx_eval_driver), every mode."""
    import json
    from helpers import build_model, tiny_config
    from utils.eval_utils import co_smoothing_core
    z, meta = load_npz("eval_driver.npz")
    N = meta["N"]
    model = build_model(tiny_config(), N, 2, seed=meta["model_seed"])
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")})
    model.cuda().train()                                   # the driver switches to eval itself and restores the mode
    regions = np.array(json.loads(bytes(z["regions"]).decode()))
    K_, T = meta["K"], meta["T"]
    batch = dict(spikes_data=torch.from_numpy(z["spikes"]).cuda(), target=torch.from_numpy(z["behavior"]).cuda(),
                 time_attn_mask=torch.ones(K_, T, dtype=torch.int64, device="cuda"),
                 spikes_timestamps=torch.arange(T, device="cuda").unsqueeze(0).repeat(K_, 1), eid=["synthetic"] * K_,
                 neuron_regions=np.asarray([regions] * K_))
    for cid, c in enumerate(meta["cases"]):
        if c["mode"] == "per_neuron":
            for j, n_i in enumerate(c["neurons"]):
                res = co_smoothing_core(model, batch, "per_neuron", heldout_idxs=[n_i], region_list=regions)
                np.testing.assert_allclose(res["rates"].cpu().numpy(), z[f"c{cid}/rates{j}"], rtol=2e-4, atol=1e-6)
                assert res["bps"][0] == pytest.approx(c["bps"][j], rel=2e-3, abs=2e-4)
            continue
        hd = c.get("held_out_list", c.get("heldout_idxs"))
        res = co_smoothing_core(model, batch, c["mode"], heldout_idxs=hd, target_regions=c.get("target_regions"), region_list=regions)
        np.testing.assert_allclose(res["rates"].cpu().numpy(), z[f"c{cid}/rates"], rtol=2e-4, atol=1e-6, err_msg=c["mode"])
        if "heldout" in c:
            assert list(res["neurons"]) == c["heldout"]
        np.testing.assert_allclose(res["bps"], np.asarray(c["bps"], dtype=np.float64), rtol=2e-3, atol=2e-4, equal_nan=True, err_msg=c["mode"])
        if not np.isnan(c["loss"]):
            assert res["loss"].item() == pytest.approx(c["loss"], rel=1e-4)
        assert res["r2"].shape == (len(res["neurons"]),)
    assert {c["mode"] for c in meta["cases"]} == {"per_neuron", "forward_pred", "inter_region", "intra_region", "modal_spike", "modal_behavior"}
    assert model.training


@pytest.mark.gpu
def test_eval_driver_at_evaluation_size_vs_reference_fixture():
    """The same driver at the size an evaluation runs at: K = 64 test trials, T = 100, N = 668 through the default model (d_model 256, 5 + 5
    layers; model and inputs regenerated from the fixture's seeds), every mode incl. modal_behavior.  Pinned: per-neuron bits/spike of
    all 668 neurons, the loss, per-neuron rate sums and a strided sample of the rates (tests/golden/eval_driver_big.npz, made by
    importing the reference: oracle/make_goldens.py:fx_eval_driver_big)."""
    from helpers import build_model, model_config
    from utils.eval_utils import co_smoothing_core
    z, meta = load_npz("eval_driver_big.npz")
    K_, T, N = meta["K"], meta["T"], meta["N"]
    g = torch.Generator().manual_seed(meta["data_seed"])              # = oracle/make_goldens.py:eval_big_inputs
    base = 0.2 + 1.6 * torch.rand(N, generator=g)
    spikes = torch.poisson(base.expand(K_, T, N).contiguous(), generator=g)
    beh = torch.randn(K_, T, 2, generator=g)
    regions = np.array(["CA1", "PO", "LP", "DG", "VISa"])[torch.randint(0, 5, (N,), generator=g).numpy()]
    np.testing.assert_array_equal(spikes.double().sum(dim=(0, 1)).numpy(), z["spikes_sum"])
    model = build_model(model_config(), N, 2, seed=meta["model_seed"]).cuda().train()
    batch = dict(spikes_data=spikes.cuda(), target=beh.cuda(), time_attn_mask=torch.ones(K_, T, dtype=torch.int64, device="cuda"),
                 spikes_timestamps=torch.arange(T, device="cuda").unsqueeze(0).repeat(K_, 1), eid=["synthetic"] * K_,
                 neuron_regions=np.asarray([regions] * K_))
    sk, st_, sn = meta["rate_stride"]

    def check(res, cid, tag, c):
        r = res["rates"].cpu().numpy()
        np.testing.assert_allclose(r[::sk, ::st_, ::sn], z[f"c{cid}/{tag}"], rtol=5e-4, atol=1e-5, err_msg=c["mode"])
        np.testing.assert_allclose(r.astype(np.float64).sum(axis=(0, 1)), z[f"c{cid}/{tag}_sum"], rtol=1e-4, err_msg=c["mode"])
    for cid, c in enumerate(meta["cases"]):
        if c["mode"] == "per_neuron":
            for j, n_i in enumerate(c["neurons"]):
                res = co_smoothing_core(model, batch, "per_neuron", heldout_idxs=[n_i], region_list=regions)
                check(res, cid, f"rates{j}", c)
                assert res["bps"][0] == pytest.approx(c["bps"][j], rel=2e-3, abs=2e-4)
            continue
        hd = c.get("held_out_list", c.get("heldout_idxs"))
        res = co_smoothing_core(model, batch, c["mode"], heldout_idxs=hd, target_regions=c.get("target_regions"), region_list=regions)
        check(res, cid, "rates", c)
        if "heldout" in c:
            assert list(res["neurons"]) == c["heldout"]
        assert len(res["bps"]) == len(c["bps"])
        np.testing.assert_allclose(res["bps"], np.asarray(c["bps"], dtype=np.float64), rtol=2e-3, atol=5e-4, equal_nan=True, err_msg=c["mode"])
        if not np.isnan(c["loss"]):
            assert res["loss"].item() == pytest.approx(c["loss"], rel=1e-4)
        assert res["r2"].shape == (len(res["neurons"]),)

"""CPU-side checks: C-ABI exports, host logic of the API mirror (config, init order, masker, parameter
layout, trainer batch building), and that the product refuses to run without a GPU (no fallback)."""
import math
import random

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz
from helpers import build_model, load_config, tiny_config


def test_library_exports_every_header_symbol():
    from multi_modal_foundation_model_amd import _lib as L
    lib = L.lib()
    syms = L.header_symbols()
    assert len(syms) >= 25
    assert [s for s in syms if not hasattr(lib, s)] == []
    assert set(syms) == set(L._PROTOS), "ctypes prototypes and include/mmfm.h must list the same functions"
    assert lib.mmfm_version() == 401
    assert lib.mmfm_last_error() is not None


def test_config_matches_reference_contract():
    cfg = load_config()
    assert cfg.model.model_class == "MultiModal"
    t = cfg.model.encoder.transformer
    assert (t.n_layers, t.hidden_size, t.n_heads, t.inter_size, t.dropout, t.act) == (5, 256, 8, 512, 0.4, "gelu")
    e = cfg.model.decoder.embedder
    assert (e.n_modality, e.n_channels, e.max_F, e.mult, e.act, e.scale, e.dropout) == (2, 668, 100, 2, "softsign", 1, 0.2)
    assert (cfg.model.masker.mode, cfg.model.masker.ratio, cfg.model.masker.force_active) == ("temporal", 0.3, True)
    assert (cfg.optimizer.lr, cfg.optimizer.wd, cfg.optimizer.eps, cfg.optimizer.warmup_pct, cfg.optimizer.div_factor) == (1e-4, 0.01, 1e-8, 0.15, 10)
    assert (cfg.training.train_batch_size, cfg.training.mask_type, cfg.seed) == (16, "embd", 42)
    from utils.config_utils import config_from_kwargs
    c = config_from_kwargs({"a.b": "[1,2]", "a.c": "null", "d": "true", "e": "-3", "f": "1e-3", "g": "txt"})
    assert c == {"a": {"b": [1, 2], "c": None}, "d": True, "e": -3, "f": 1e-3, "g": "txt"}
    assert c.a.b == [1, 2]


def test_init_order_bit_exact_vs_reference():
    g = load_json("init_order.json")
    model = build_model(load_config().model, 668, 2, seed=42)
    sd = model.state_dict()
    assert list(sd.keys()) == [e["key"] for e in g["state_dict"]]
    assert [k for k, _ in model.named_parameters()] == g["named_parameters"]
    assert sum(p.numel() for p in model.parameters()) == 9409126
    for e in g["state_dict"]:
        v = sd[e["key"]]
        assert list(v.shape) == e["shape"]
        assert [float(x) for x in v.flatten()[:4]] == e["first"], e["key"]
        assert float(v.double().sum()) == pytest.approx(e["sum"], rel=1e-12, abs=1e-12), e["key"]
    for m in ("ap", "behavior"):
        assert model.decoder_embeddings[m].embedder.mod_emb.weight is model.encoder_embeddings[m].embedder.mod_emb.weight


def test_masker_bit_exact_vs_reference():
    from models.masker import Masker
    from utils.config_utils import DictConfig
    z, cases = load_npz("masker_bits.npz")
    for c in cases:
        i = c["id"]
        mk = Masker(DictConfig(c["cfg"]))
        mk.train()
        torch.manual_seed(c["seed"])
        _, m_ap = mk(torch.from_numpy(z[f"c{i}/ap"]).clone(), np.full((4, 9), "XX"))
        _, m_bh = mk(torch.from_numpy(z[f"c{i}/bh"]).clone(), None)
        np.testing.assert_array_equal(m_ap.numpy(), z[f"c{i}/mask_ap"])
        np.testing.assert_array_equal(m_bh.numpy(), z[f"c{i}/mask_bh"])
        np.testing.assert_array_equal(torch.rand(3).numpy(), z[f"c{i}/after"])
    mk = Masker(DictConfig(dict(cases[0]["cfg"], ratio=0)))
    _, zz = mk(torch.ones(2, 3, 4), None)
    np.testing.assert_array_equal(zz.numpy(), z["zero_ratio_mask"])
    # token_mask_only keeps the token-level draw (first two generator calls) identical
    c = cases[0]
    a, b = Masker(DictConfig(c["cfg"])), Masker(DictConfig(c["cfg"]))
    x = torch.from_numpy(z["c0/ap"])
    torch.manual_seed(3); xa, ma = a(x.clone(), np.full((4, 9), "XX"))
    torch.manual_seed(3); xb, mb = b(x.clone(), np.full((4, 9), "XX"), token_mask_only=True)
    assert torch.equal(ma, mb)
    assert not torch.equal(xa, x) and torch.equal(xb, x)       # the corruption draws are what the switch skips
    # the switch is a per-call argument: the attribute alone (what a trainer sets) changes nothing for callers that USE the spikes,
    # and it never survives a pickle (model_best.pt must not carry one run's throughput switch)
    b.token_mask_only = True
    torch.manual_seed(3); xc, mc = b(x.clone(), np.full((4, 9), "XX"))
    assert torch.equal(xc, xa) and torch.equal(mc, ma)
    import pickle
    assert pickle.loads(pickle.dumps(b)).token_mask_only is False


def test_param_layout_covers_parameters_contiguously():
    from multi_modal_foundation_model_amd.engine import EngineConfig, ParamLayout
    from multi_modal_foundation_model_amd.ddp import GradBuckets, backward_order
    cfg = load_config().model
    model = build_model(cfg, 668, 2, seed=0)
    ec = EngineConfig.from_model_config(cfg, [("ap", 668), ("behavior", 2)])
    lay = ParamLayout(ec)
    named = dict(model.named_parameters())
    assert set(lay.entries) == set(named)
    spans = sorted((off, off + int(np.prod(shape)), n) for n, (off, shape) in lay.entries.items())
    for (a0, a1, _), (b0, b1, _) in zip(spans, spans[1:]):
        assert a1 <= b0, "parameters overlap in the flat buffer"
    for n, (off, shape) in lay.entries.items():
        assert tuple(named[n].shape) == shape
    H = 256
    for p in ("encoder.0.attn", "decoder.3.attn"):
        q, k, v = (lay.entries[f"{p}.{x}.weight"][0] for x in ("query", "key", "value"))
        assert (k - q, v - k) == (H * H, H * H) and lay.alias[f"{p}.qkv.weight"] == (q, (3 * H, H))
    kk, vv = (lay.entries[f"decoder.1.cross_attn.{x}.bias"][0] for x in ("key", "value"))
    assert vv - kk == H and lay.alias["decoder.1.cross_attn.kv.bias"] == (kk, (2 * H,))
    # segments tile [0, n) in forward order; DDP buckets are contiguous ranges in backward order
    segs = lay.segments
    assert segs[0][1] == 0 and all(a[2] == b[1] for a, b in zip(segs, segs[1:]))
    bk = GradBuckets(lay, ec, bucket_bytes=8 << 20).buckets
    assert bk[0][2] == segs[-1][2] and bk[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(bk, bk[1:]))
    assert [b[0] for b in bk][-1] == "embed" and backward_order(lay, ec)[0] == "head"


def test_product_refuses_cpu():
    model = build_model(tiny_config(), 12, 2, seed=0)
    from oracle.mm_oracle import make_mod_dict, synth_batch
    md = make_mod_dict(synth_batch(2, 8, 12, 2, seed=0), "encoding")
    for d in md.values():
        d["targets_modality"] = d["inputs_modality"]
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        model(md)


def test_trainer_selects_masker_stream_from_config():
    """the out-of-the-box trainer must not be bound by the masker's dead [B,T,N] host draws.  mask_type 'embd' (the
    shipped trainer_mm.yaml) switches the model's masker to its token-mask-only stream; training.exact_masker_stream (or
    MMFM_EXACT_MASKER=1) keeps the reference's generator walk, and so does any other mask_type."""
    from trainer.make import make_multimodal_trainer

    class Acc:
        device = torch.device("cpu")

    def make(**training):
        cfg = load_config()
        cfg["training"].update(training)
        model = build_model(tiny_config(), 12, 2, seed=1)
        make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=None, log_dir="/tmp", accelerator=Acc(),
                                lr_scheduler=None, avail_mod=["ap", "behavior"], config=cfg,
                                modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]), mixed_training=True, num_neurons=[12])
        return model.masker.token_mask_only
    assert load_config().training.mask_type == "embd"
    assert make() is True
    assert make(exact_masker_stream=True) is False
    assert make(mask_type="input") is False


def test_trainer_builds_reference_mod_dict():
    from trainer.make import make_multimodal_trainer
    from multi_modal_foundation_model_amd.synthetic import synth_batch
    seen = {}

    class Stub(torch.nn.Module):
        def forward(self, md):
            seen["md"] = md
            return "out"

    class Acc:
        device = torch.device("cpu")
    cfg = load_config()
    tr = make_multimodal_trainer(model=Stub(), train_dataloader=[], eval_dataloader=[], optimizer=None, log_dir="/tmp",
                                 accelerator=Acc(), lr_scheduler=None, avail_mod=["ap", "behavior"], config=cfg,
                                 modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]), mixed_training=True,
                                 num_neurons=[12])
    batch = synth_batch(3, 8, 12, 2, seed=1)
    assert tr._forward_model_outputs(dict(batch), None, "encoding") == "out"
    md = seen["md"]
    assert list(md) == ["ap", "behavior"]
    assert md["ap"]["eval_mask"].shape == (3, 8, 12) and int(md["ap"]["eval_mask"].min()) == 1
    assert md["behavior"]["eval_mask"].shape == (3, 8, 12) and int(md["behavior"]["eval_mask"].max()) == 0   # shaped like spikes (upstream quirk)
    assert md["ap"]["inputs_regions"].shape == (3, 12) and md["behavior"]["inputs"].shape == (3, 8, 2)
    tr._forward_model_outputs(dict(batch), None, "decoding")
    assert seen["md"]["ap"]["eval_mask"].shape == (3, 8, 2) and int(seen["md"]["behavior"]["eval_mask"].min()) == 1
    tr._forward_model_outputs(dict(batch), None, "token_masking")
    assert seen["md"]["ap"]["eval_mask"] is None
    with pytest.raises(Exception, match="Training objective not implemented yet"):
        tr._forward_model_outputs(dict(batch), None, None)
    g = load_json("trainer_io.json")
    random.seed(42)
    assert [random.sample(tr.training_schemes, 1)[0] for _ in range(6)] == g["objectives"]


def test_context_mask_helper():
    from multi_modal.mm_utils import create_context_mask
    m = create_context_mask(0, -1, 5)
    assert torch.equal(m, torch.tril(torch.ones(5, 5, dtype=torch.int64)))
    assert torch.equal(create_context_mask(-1, -1, 4), torch.ones(4, 4, dtype=torch.int64))


def test_lazy_regions_behaves_like_the_region_array():
    """trainer/base.py:57 builds np.asarray(batch['neuron_regions']).T every step; the mirror defers it (34 ms of host time
    at B = 1024) behind an object that reads like that array."""
    from trainer.base import LazyRegions
    raw = [["CA1", "CA1", "PO"], ["LP", "LP", "LP"], ["PO", "CA1", "PO"], ["nan", "nan", "nan"]]      # N = 4 lists of B = 3 strings
    want = np.asarray(raw).T
    lz = LazyRegions(raw)
    assert lz._arr is None                                   # nothing converted yet
    np.testing.assert_array_equal(np.asarray(lz), want)
    assert lz.shape == (3, 4) and len(lz) == 3
    np.testing.assert_array_equal(np.unique(lz), np.unique(want))
    np.testing.assert_array_equal(lz == "PO", want == "PO")
    np.testing.assert_array_equal(lz != "PO", want != "PO")
    np.testing.assert_array_equal(lz[1], want[1])
    np.testing.assert_array_equal(lz.T, want.T)
    assert [list(r) for r in lz] == [list(r) for r in want]

"""Builders shared by bench.py, scripts/ and the tests: the API mirror's model exactly as train_multi_modal.py builds it
(train_multi_modal.py:160-210: construction order = RNG contract)."""
import copy
import os

import torch

import sys

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "src")
if SRC not in sys.path:
    sys.path.insert(0, SRC)


def load_config():
    from utils.config_utils import config_from_kwargs, update_config
    cwd = os.getcwd()
    os.chdir(os.path.dirname(SRC))          # config paths are 'src/configs/...' relative, like the reference
    try:
        cfg = config_from_kwargs({"model": "include:src/configs/multi_modal/mm.yaml"})
        cfg = update_config("src/configs/multi_modal/trainer_mm.yaml", cfg)
    finally:
        os.chdir(cwd)
    return cfg


def model_config(H=None, heads=None, inter=None, n_enc=None, n_dec=None, max_F=None, dropout=None, emb_dropout=None,
                 sep=None, causal=None, n_modality=None):
    from utils.config_utils import DictConfig
    m = copy.deepcopy(dict(load_config()["model"]))
    for side in ("encoder", "decoder"):
        e, t = m[side]["embedder"], m[side]["transformer"]
        if max_F is not None: e["max_F"] = max_F
        if n_modality is not None: e["n_modality"] = n_modality
        if emb_dropout is not None: e["dropout"] = emb_dropout
        if H is not None: t["hidden_size"] = H
        if heads is not None: t["n_heads"] = heads
        if inter is not None: t["inter_size"] = inter
        if dropout is not None: t["dropout"] = dropout
    if n_enc is not None: m["encoder"]["transformer"]["n_layers"] = n_enc
    if n_dec is not None: m["decoder"]["transformer"]["n_layers"] = n_dec
    if sep is not None: m["decoder"]["decoder_sep_mask"] = sep
    if causal is not None: m["decoder"]["decoder_causal_mask"] = causal
    return DictConfig(m)


def tiny_config(**kw):
    base = dict(H=32, heads=4, inter=64, n_enc=1, n_dec=1, max_F=8, dropout=0.0, emb_dropout=0.0)
    base.update(kw)
    return model_config(**base)


def build_model(mcfg, n_ap, n_beh, seed=None):
    """train_multi_modal.py:160-189 (construction order = RNG contract)."""
    from multi_modal.mm import MultiModal
    from multi_modal.encoder_embeddings import EncoderEmbedding
    from multi_modal.decoder_embeddings import DecoderEmbedding
    if seed is not None:
        torch.manual_seed(seed)
    enc, dec = {}, {}
    for mod in ("ap", "behavior"):
        enc[mod] = EncoderEmbedding(hidden_size=mcfg.encoder.transformer.hidden_size, n_channel=n_ap if mod == "ap" else n_beh,
                                    config=mcfg.encoder)
    for mod in ("ap", "behavior"):
        dec[mod] = DecoderEmbedding(hidden_size=mcfg.decoder.transformer.hidden_size, n_channel=n_ap if mod == "ap" else n_beh,
                                    output_channel=n_ap if mod == "ap" else n_beh, config=mcfg.decoder)
    return MultiModal(enc, dec, avail_mod=["ap", "behavior"], config=mcfg, share_modality_embeddings=True)


def build_model_mods(mcfg, mods, seed=None):
    """Same construction order for an arbitrary modality list [(name, channels)] (BASELINE configs[4]: 3 modalities)."""
    from multi_modal.mm import MultiModal
    from multi_modal.encoder_embeddings import EncoderEmbedding
    from multi_modal.decoder_embeddings import DecoderEmbedding
    if seed is not None:
        torch.manual_seed(seed)
    H = mcfg.encoder.transformer.hidden_size
    enc = {m: EncoderEmbedding(hidden_size=H, n_channel=n, config=mcfg.encoder) for m, n in mods}
    dec = {m: DecoderEmbedding(hidden_size=H, n_channel=n, output_channel=n, config=mcfg.decoder) for m, n in mods}
    return MultiModal(enc, dec, avail_mod=[m for m, _ in mods], config=mcfg, share_modality_embeddings=True)


def make_optimizer(model, total_steps, lr=1e-4, wd=0.01, eps=1e-8):
    from torch.optim.lr_scheduler import OneCycleLR
    from multi_modal_foundation_model_amd.optim import make_optimizer as mk
    opt = mk(model, lr=lr, weight_decay=wd, eps=eps)
    sch = OneCycleLR(optimizer=opt, total_steps=total_steps, max_lr=lr, pct_start=0.15, div_factor=10)
    return opt, sch

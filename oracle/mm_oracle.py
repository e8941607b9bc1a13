"""CPU oracle: a functional restatement of the reference's masked-pretraining path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is shipped or measured as the
product: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it, and only as the checker / the CPU baseline.  The product path
(`multi_modal_foundation_model_amd/`) never falls back to this file.

Parity status: PINNED.  Every function below is checked in
`tests/test_oracle_golden.py` against fixtures produced by importing the
reference itself (`oracle/make_goldens.py` -> `tests/golden/`).

Plain torch (fp32) ops on whatever device the inputs live on; parameters are a
flat ``{state_dict key: tensor}`` mapping whose keys equal the reference's
``MultiModal.state_dict()`` keys.  Citations are to /root/reference/src/.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MODS_DEFAULT = ("ap", "behavior")


@dataclass
class OracleCfg:
    """The subset of configs/multi_modal/mm.yaml the path reads."""
    hidden: int = 256
    heads: int = 8
    inter: int = 512
    n_enc: int = 5
    n_dec: int = 5
    max_F: int = 100
    mult: int = 2
    n_modality: int = 2
    embed_scale: Optional[float] = 1.0     # None -> hidden**0.5  (encoder_embeddings.py:34)
    embed_dropout: float = 0.2
    dropout: float = 0.4
    sep_mask: bool = False                 # mm.yaml:54
    causal_mask: bool = False              # mm.yaml:55
    fixup: bool = True
    avail_mod: Tuple[str, ...] = MODS_DEFAULT
    channels: Dict[str, int] = field(default_factory=lambda: {"ap": 668, "behavior": 2})

    @staticmethod
    def from_model_config(mc, channels, avail_mod=MODS_DEFAULT) -> "OracleCfg":
        et, ee = mc["encoder"]["transformer"], mc["encoder"]["embedder"]
        return OracleCfg(hidden=et["hidden_size"], heads=et["n_heads"], inter=et["inter_size"],
                         n_enc=et["n_layers"], n_dec=mc["decoder"]["transformer"]["n_layers"],
                         max_F=ee["max_F"], mult=ee["mult"], n_modality=ee["n_modality"],
                         embed_scale=ee["scale"], embed_dropout=ee["dropout"], dropout=et["dropout"],
                         sep_mask=mc["decoder"]["decoder_sep_mask"],
                         causal_mask=mc["decoder"]["decoder_causal_mask"], fixup=et["fixup_init"],
                         avail_mod=tuple(avail_mod), channels=dict(channels))


# ----------------------------------------------------------------------------- init
def init_state_dict(cfg: OracleCfg, seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Parameter creation in the reference's RNG order (train_multi_modal.py:160-189).

    Embedder: token_embed, projection, mod_emb, pos_embed (encoder_embeddings.py:28-40);
    decoder embedder additionally `out` (decoder_embeddings.py:81-83); each block
    ln1, attn(q,k,v,out_proj), [cross_attn, query_norm, context_norm], ln2, mlp(up,down)
    then the fixup rescale (encoder_embeddings.py:118-129).  The decoder embedder's own
    mod_emb is drawn and then replaced by the encoder's (mm.py:84-87).
    """
    if seed is not None:
        torch.manual_seed(seed)
    nn = torch.nn
    H, I = cfg.hidden, cfg.inter
    sd: Dict[str, torch.Tensor] = {}

    def lin(prefix, i, o):
        l = nn.Linear(i, o)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = l.weight.data, l.bias.data

    def emb(prefix, n, d):
        sd[prefix + ".weight"] = nn.Embedding(n, d).weight.data

    def ln(prefix):
        sd[prefix + ".weight"], sd[prefix + ".bias"] = torch.ones(H), torch.zeros(H)

    for side in ("encoder", "decoder"):
        for mod in cfg.avail_mod:
            n = cfg.channels[mod]
            p = f"{side}_embeddings.{mod}.embedder"
            lin(p + ".token_embed", n, n * cfg.mult)
            lin(p + ".projection", n * cfg.mult, H)
            emb(p + ".mod_emb", cfg.n_modality, H)
            emb(p + ".pos_embed", cfg.max_F, H)
            if side == "decoder":
                lin(f"decoder_embeddings.{mod}.out", H, n)

    def attn(prefix):
        for nm in ("query", "key", "value", "out_proj"):
            lin(f"{prefix}.{nm}", H, H)

    def fix(prefix, names, n_layers):
        c = 0.67 * n_layers ** (-1.0 / 4.0)
        for nm in names:
            k = f"{prefix}.{nm}"
            if k.endswith("_proj.weight"):
                sd[k] = c * sd[k]
            elif k.endswith("value.weight"):
                sd[k] = c * (sd[k] * (2 ** 0.5))

    for i in range(cfg.n_enc):
        p = f"encoder.{i}"
        ln(p + ".ln1"); attn(p + ".attn"); ln(p + ".ln2")
        lin(p + ".mlp.up_proj", H, I); lin(p + ".mlp.down_proj", I, H)
        if cfg.fixup:
            fix(p, ["attn.value.weight", "attn.out_proj.weight", "mlp.up_proj.weight",
                    "mlp.down_proj.weight"], cfg.n_enc)
    ln("encoder_norm")
    lin("decoder_proj_context", H, H)
    for i in range(cfg.n_dec):
        p = f"decoder.{i}"
        ln(p + ".ln1"); attn(p + ".attn"); attn(p + ".cross_attn")
        ln(p + ".query_norm"); ln(p + ".context_norm"); ln(p + ".ln2")
        lin(p + ".mlp.up_proj", H, I); lin(p + ".mlp.down_proj", I, H)
        if cfg.fixup:
            fix(p, ["attn.value.weight", "attn.out_proj.weight", "cross_attn.value.weight",
                    "cross_attn.out_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"],
                cfg.n_dec)
    ln("decoder_norm")
    # share_modality_embeddings (mm.py:84-87): decoder mod_emb IS the encoder's tensor
    for mod in cfg.avail_mod:
        sd[f"decoder_embeddings.{mod}.embedder.mod_emb.weight"] = \
            sd[f"encoder_embeddings.{mod}.embedder.mod_emb.weight"]
    return state_dict_order(sd, cfg)


def state_dict_order(sd, cfg: OracleCfg):
    """Key order of the reference's state_dict(): module registration order (mm.py:57-77)."""
    keys: List[str] = []
    for side in ("encoder", "decoder"):
        for mod in cfg.avail_mod:
            p = f"{side}_embeddings.{mod}.embedder"
            keys += [p + ".token_embed.weight", p + ".token_embed.bias", p + ".projection.weight",
                     p + ".projection.bias", p + ".mod_emb.weight", p + ".pos_embed.weight"]
            if side == "decoder":
                keys += [f"decoder_embeddings.{mod}.out.weight", f"decoder_embeddings.{mod}.out.bias"]

    def lnk(p):
        return [p + ".weight", p + ".bias"]

    def attnk(p):
        return sum(([f"{p}.{n}.weight", f"{p}.{n}.bias"] for n in ("query", "key", "value", "out_proj")), [])

    def mlpk(p):
        return [p + ".up_proj.weight", p + ".up_proj.bias", p + ".down_proj.weight", p + ".down_proj.bias"]
    for i in range(cfg.n_enc):
        p = f"encoder.{i}"
        keys += lnk(p + ".ln1") + attnk(p + ".attn") + lnk(p + ".ln2") + mlpk(p + ".mlp")
    keys += lnk("encoder_norm") + lnk("decoder_proj_context")
    for i in range(cfg.n_dec):
        p = f"decoder.{i}"
        keys += (lnk(p + ".ln1") + attnk(p + ".attn") + attnk(p + ".cross_attn") + lnk(p + ".query_norm")
                 + lnk(p + ".context_norm") + lnk(p + ".ln2") + mlpk(p + ".mlp"))
    keys += lnk("decoder_norm")
    assert set(keys) == set(sd.keys()), set(keys) ^ set(sd.keys())
    return {k: sd[k] for k in keys}


def share_mod_emb(sd, cfg: OracleCfg):
    """Re-establish share_modality_embeddings (mm.py:84-87) after a state dict was cloned/loaded:
    the decoder tokenisers' mod_emb must be the SAME tensor object as the encoder's."""
    for m in cfg.avail_mod:
        sd[f"decoder_embeddings.{m}.embedder.mod_emb.weight"] = sd[f"encoder_embeddings.{m}.embedder.mod_emb.weight"]
    return sd


def trainable_keys(sd, cfg: OracleCfg) -> List[str]:
    """named_parameters() de-duplicates the shared mod_emb."""
    drop = {f"decoder_embeddings.{m}.embedder.mod_emb.weight" for m in cfg.avail_mod}
    return [k for k in sd if k not in drop]


# ----------------------------------------------------------------------------- masker
class OracleMasker:
    """Restatement of models/masker.py:56-168 for the modes the model allows
    (`temporal`, mm.py:68) plus `random`/`neuron`/`co-smooth`/`forward-pred`.

    CPU-generator call order in temporal mode (masker.py:81,86/92,132,158,160,161):
    bernoulli(expand_prob) [, randint(timespan)], bernoulli([B,T]), bernoulli([B,T,N]),
    bernoulli([B,T,N]), rand([B,T,N]).  All drawn from the CPU generator.
    """

    def __init__(self, mcfg: dict):
        self.force_active = mcfg.get("force_active", False)
        self.mode = mcfg["mode"]
        self.ratio = mcfg["ratio"]
        self.zero_ratio = mcfg["zero_ratio"]
        self.random_ratio = mcfg["random_ratio"]
        self.expand_prob = mcfg["expand_prob"]
        self.max_timespan = mcfg["max_timespan"]
        self.channels = mcfg["channels"]
        self.timesteps = mcfg["timesteps"]
        self.mask_regions = mcfg["mask_regions"]
        self.target_regions = mcfg["target_regions"]
        self.training = True

    def __call__(self, spikes: torch.Tensor, regions=None):
        if (not self.training and not self.force_active) or self.target_regions is None \
                or self.mask_regions is None or self.ratio == 0:            # masker.py:62-69
            return spikes, torch.zeros_like(spikes).to(torch.int64)
        if "all" in self.mask_regions:                                      # masker.py:72-76
            self.mask_regions = list(np.unique(regions))
        if "all" in self.target_regions:
            self.target_regions = list(np.unique(regions))
        B, T, N = spikes.shape
        ratio = self.ratio
        timespan = 1
        if self.mode == "temporal":
            if torch.bernoulli(torch.tensor(self.expand_prob).float()):     # masker.py:81
                timespan = int(torch.randint(1, self.max_timespan + 1, (1,)).item())
            probs = torch.full((B, T), ratio / timespan)
        elif self.mode == "neuron":
            probs = torch.full((B, N), ratio)
        elif self.mode == "random":
            probs = torch.full((B, T, N), ratio)
        elif self.mode == "co-smooth":
            probs = torch.zeros(N)
            probs[list(self.channels)] = 1
        elif self.mode == "forward-pred":
            probs = torch.zeros(T)
            probs[list(self.timesteps)] = 1
        else:
            raise Exception(f"Masking mode {self.mode} not implemented")
        mask = torch.bernoulli(probs)                                       # masker.py:132
        if self.mode == "temporal":
            if timespan > 1:                                                # masker.py:170-174
                k = torch.ones(1, 1, timespan)
                mask = (F.conv1d(mask[:, None], k, padding="same")[:, 0] >= 1)
            mask = mask[:, :, None].expand(B, T, N).bool()
        elif self.mode == "neuron":
            mask = mask[:, None, :].expand(B, T, N).bool()
        elif self.mode == "co-smooth":
            mask = mask[None, None, :].expand(B, T, N).bool()
        elif self.mode == "forward-pred":
            mask = mask[None, :, None].expand(B, T, N).bool()
        else:
            mask = mask.bool()
        dev = spikes.device
        mask = mask.to(dev)
        zero_idx = torch.bernoulli(torch.full((B, T, N), float(self.zero_ratio))).to(dev).bool() & mask
        spikes[zero_idx] = 0
        rnd_idx = torch.bernoulli(torch.full((B, T, N), float(self.random_ratio))).to(dev).bool() & mask & ~zero_idx
        rnd = (spikes.max() * torch.rand((B, T, N)).to(dev)).to(spikes.dtype)   # CPU generator, see class doc
        spikes[rnd_idx] = rnd[rnd_idx]
        return spikes, mask.to(torch.int64)


# ----------------------------------------------------------------------------- model pieces
def softsign(x):
    return x / (1 + x.abs())


def embed(sd, p, inputs, ts, mod_idx, cfg: OracleCfg, training, gen=None):
    """EncoderEmbeddingLayer.forward / DecoderEmbeddingLayer.forward
    (encoder_embeddings.py:44-61, decoder_embeddings.py:43-61)."""
    scale = cfg.hidden ** 0.5 if cfg.embed_scale is None else cfg.embed_scale
    x = F.linear(inputs, sd[p + ".token_embed.weight"], sd[p + ".token_embed.bias"])
    x = softsign(x) * scale
    x = F.linear(x, sd[p + ".projection.weight"], sd[p + ".projection.bias"])
    B, T, _ = inputs.shape
    e = sd[p + ".mod_emb.weight"][mod_idx][None, None, :].expand(B, T, -1).clone()
    e = e + sd[p + ".pos_embed.weight"][ts]
    return F.dropout(x, cfg.embed_dropout, training), e


def zero_masked_tokens(tokens, mask):
    """mm.py:147-149 / :169-171 — indices come from SAMPLE 0's mask, applied to every b."""
    ids = torch.argwhere(mask[0] == 1).squeeze()
    tokens = tokens.clone()
    tokens[:, ids, :] = 0.0
    return tokens


def encoder_attn_mask(keypad):
    """mm.py:152-158: eye | (ones & keypad[b,k])  -> bool [B,L,L]."""
    B, L = keypad.shape
    eye = torch.eye(L, device=keypad.device, dtype=torch.int64).expand(B, L, L)
    return (eye | keypad[:, None, :].expand(B, L, L)).bool()


def decoder_attn_mask(keypad, mod_mask, causal, sep):
    """mm.py:178-194 (create_context_mask(0,-1,N) is lower-triangular incl. diagonal,
    mm_utils.py:17-28)."""
    B, L = keypad.shape
    if causal:
        m = torch.tril(torch.ones(L, L, dtype=torch.int64, device=keypad.device))[None].expand(B, L, L)
    else:
        m = keypad[:, None, :].expand(B, L, L)
    m = m.bool()
    if sep:
        m = m | (mod_mask[:, None, :] != mod_mask[:, :, None])
    return m


def attention(sd, p, xq, xkv, mask, heads, drop, training):
    """Attention.forward / CrossAttention.forward (mm_utils.py:97-114, 139-152)."""
    B, Lq, H = xq.shape
    Lk = xkv.shape[1]
    dh = H // heads
    q = F.linear(xq, sd[p + ".query.weight"], sd[p + ".query.bias"]).view(B, Lq, heads, dh).transpose(1, 2)
    k = F.linear(xkv, sd[p + ".key.weight"], sd[p + ".key.bias"]).view(B, Lk, heads, dh).transpose(1, 2)
    v = F.linear(xkv, sd[p + ".value.weight"], sd[p + ".value.bias"]).view(B, Lk, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    s = s.masked_fill(~mask[:, None, :, :], float("-inf"))
    a = F.dropout(torch.softmax(s, dim=-1), drop, training)
    o = (a @ v).transpose(1, 2).contiguous().view(B, Lq, H)
    return F.linear(F.dropout(o, drop, training), sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def mlp(sd, p, x, drop, training):
    """MLP.forward (mm_utils.py:50-52); ACT2FN['gelu'] is the exact-erf GELU."""
    h = F.gelu(F.linear(x, sd[p + ".up_proj.weight"], sd[p + ".up_proj.bias"]))
    return F.dropout(F.linear(h, sd[p + ".down_proj.weight"], sd[p + ".down_proj.bias"]), drop, training)


def forward(sd, mod_dict, cfg: OracleCfg, training: bool = False, masker: Optional[OracleMasker] = None,
            keep: bool = False):
    """MultiModal.forward (mm.py:242-308).  `mod_dict` as built by
    MultiModalTrainer._forward_model_outputs (trainer/base.py:51-103).  Returns a dict."""
    mods = [m for m in mod_dict]
    mod_to_idx = {m: i for i, m in enumerate(cfg.avail_mod)}
    dp = cfg.dropout
    masks, x_e, e_e, x_d, e_d, keyp, modm = {}, [], [], [], [], [], []
    for mod in mods:
        d = mod_dict[mod]
        inp, tgt = d["inputs"], d["targets"]
        if mod == "behavior" and inp.dim() == 2:                       # mm.py:248-250
            inp, tgt = inp.unsqueeze(-1), tgt.unsqueeze(-1)
            d["inputs"], d["targets"] = inp, tgt
        if d.get("masking_mode"):
            raise UnboundLocalError("mask")                            # mm.py:256-263 vs :272 (upstream bug)
        if d["eval_mask"] is None:
            _, mk = masker(inp.clone(), d.get("inputs_regions") if mod == "ap" else None)   # mm.py:267
        else:
            mk = d["eval_mask"]
        masks[mod] = mk[:, :, 0] & d["inputs_attn_mask"]              # mm.py:270
    for mod in mods:                                                   # mm.py:277-279
        d = mod_dict[mod]
        x, e = embed(sd, f"encoder_embeddings.{mod}.embedder", d["inputs"], d["inputs_timestamp"],
                     mod_to_idx[mod], cfg, training)
        x_e.append(x); e_e.append(e)
        keyp.append(d["inputs_attn_mask"])
        modm.append(torch.full_like(masks[mod], mod_to_idx[mod], dtype=torch.int16))
    for mod in mods:                                                   # mm.py:283-285
        d = mod_dict[mod]
        x, e = embed(sd, f"decoder_embeddings.{mod}.embedder", d["inputs"], d["inputs_timestamp"],
                     mod_to_idx[mod], cfg, training)
        x_d.append(x); e_d.append(e)
    tok_mask = torch.cat([masks[m] for m in mods], 1)
    keypad = torch.cat(keyp, 1)
    mod_mask = torch.cat(modm, 1)
    enc_tokens = zero_masked_tokens(torch.cat(x_e, 1), tok_mask)       # mm.py:281
    enc_emb = torch.cat(e_e, 1)
    dec_tokens = zero_masked_tokens(torch.cat(x_d, 1), tok_mask)       # mm.py:287
    dec_emb = torch.cat(e_d, 1)
    am_enc = encoder_attn_mask(keypad)
    am_dec = decoder_attn_mask(keypad, mod_mask, cfg.causal_mask, cfg.sep_mask)

    x = enc_tokens + enc_emb                                           # mm.py:289
    for i in range(cfg.n_enc):                                         # encoder_embeddings.py:106-116
        p = f"encoder.{i}"
        h = layer_norm(sd, p + ".ln1", x)
        x = x + attention(sd, p + ".attn", h, h, am_enc, cfg.heads, dp, training)
        x = x + mlp(sd, p + ".mlp", layer_norm(sd, p + ".ln2", x), dp, training)
    enc_out = layer_norm(sd, "encoder_norm", x)
    ctx_proj = F.linear(enc_out, sd["decoder_proj_context.weight"], sd["decoder_proj_context.bias"])
    context = ctx_proj + enc_emb                                       # mm.py:292
    y = dec_tokens + dec_emb
    for i in range(cfg.n_dec):                                         # decoder_embeddings.py:133-147
        p = f"decoder.{i}"
        h = layer_norm(sd, p + ".ln1", y)
        y = y + attention(sd, p + ".attn", h, h, am_dec, cfg.heads, dp, training)
        y = y + attention(sd, p + ".cross_attn", layer_norm(sd, p + ".query_norm", y),
                          layer_norm(sd, p + ".context_norm", context), am_enc, cfg.heads, dp, training)
        y = y + mlp(sd, p + ".mlp", layer_norm(sd, p + ".ln2", y), dp, training)
    dec_out = layer_norm(sd, "decoder_norm", y)

    B = dec_out.shape[0]
    mod_loss, mod_n, preds, targets = {}, {}, {}, {}
    for mod in mods:                                                   # decoder_embeddings.py:95-109
        ym = dec_out[mod_mask == mod_to_idx[mod]]
        pr = F.linear(ym, sd[f"decoder_embeddings.{mod}.out.weight"], sd[f"decoder_embeddings.{mod}.out.bias"])
        pr = pr.reshape(B, -1, pr.shape[-1])
        tg = mod_dict[mod]["targets"]
        mk = masks[mod].unsqueeze(-1).expand_as(tg)                    # mm.py:221-233
        if mod == "ap":
            el = torch.exp(pr) - tg * pr                               # PoissonNLLLoss(log_input=True), mm.py:80
        else:
            el = (pr - tg) ** 2                                        # MSELoss, mm.py:81
        mod_loss[mod] = (el * mk).sum()
        mod_n[mod] = mk.sum()
        preds[mod], targets[mod] = pr, tg
    loss = sum(mod_loss.values()) / sum(mod_n.values())               # mm.py:237
    out = dict(loss=loss, mod_loss=mod_loss, mod_n_examples=mod_n, mod_preds=preds, mod_targets=targets,
               masks=masks)
    if keep:
        out.update(enc_out=enc_out, ctx_proj=ctx_proj, dec_out=dec_out, enc_x=x_e, enc_emb=e_e, dec_x=x_d,
                   dec_emb=e_d, am_enc=am_enc, am_dec=am_dec, mod_mask=mod_mask, enc_tokens=enc_tokens,
                   dec_tokens=dec_tokens)
    return out


# ----------------------------------------------------------------------------- optimiser
def onecycle(step: int, total_steps: int, max_lr=1e-4, pct_start=0.15, div_factor=10.0,
             final_div_factor=1e4, base_momentum=0.85, max_momentum=0.95):
    """(lr, beta1) that torch's OneCycleLR(cos, cycle_momentum) has set BEFORE optimiser
    step number `step` (0-based) — train_multi_modal.py:204-210, trainer/base.py:196-197."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    e1 = float(pct_start * total_steps) - 1
    e2 = total_steps - 1

    def cos(a, b, pct):                      # torch's _annealing_cos, same operation order
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)
    if step <= e1:
        pct = step / e1
        return cos(initial_lr, max_lr, pct), cos(max_momentum, base_momentum, pct)
    pct = (step - e1) / (e2 - e1)
    return cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)


def adamw_step(p, g, m, v, step, lr, beta1, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update, in place; `step` is 1-based."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------- data / trainer glue
def synth_batch(B, T, n_ap, n_beh, seed, pad=None, ts_shift=None):
    """SURVEY.md §8d synthetic recipe (host generator, Poisson(0.3) spikes, N(0,1) behaviour)."""
    g = torch.Generator().manual_seed(seed)
    spikes = torch.poisson(torch.full((B, T, n_ap), 0.3), generator=g)
    beh = torch.randn(B, T, n_beh, generator=g)
    attn = torch.ones(B, T, dtype=torch.int64)
    if pad is not None:
        for b, pb in enumerate(pad):
            if pb:
                attn[b, T - pb:] = 0
    ts = torch.arange(T, dtype=torch.int64)[None].repeat(B, 1)
    if ts_shift is not None:
        ts = (ts + torch.tensor(ts_shift)[:, None]) % T
    return dict(spikes_data=spikes, target=beh, time_attn_mask=attn, spikes_timestamps=ts)


def make_mod_dict(batch, objective, avail_mod=MODS_DEFAULT):
    """MultiModalTrainer._forward_model_outputs (trainer/base.py:51-103), eval_mask quirks
    included: the 'other' modality's zeros mask is shaped like the *selected* one's data."""
    spikes, beh = batch["spikes_data"], batch["target"]
    B, T, n_ap = spikes.shape
    md = {}
    for i, mod in enumerate(avail_mod):
        x = spikes if mod == "ap" else beh
        md[mod] = dict(inputs_modality=torch.tensor(i), inputs_attn_mask=batch["time_attn_mask"],
                       inputs_timestamp=batch["spikes_timestamps"], masking_mode=None,
                       inputs=x.clone(), targets=x.clone())
        if mod == "ap":
            md[mod]["inputs_regions"] = np.full((B, n_ap), "XX")
    if objective == "encoding":
        md["ap"]["eval_mask"] = torch.ones_like(spikes).to(torch.int64)
        md["behavior"]["eval_mask"] = torch.zeros_like(spikes).to(torch.int64)
    elif objective == "decoding":
        md["behavior"]["eval_mask"] = torch.ones_like(beh).to(torch.int64)
        md["ap"]["eval_mask"] = torch.zeros_like(beh).to(torch.int64)
    elif objective == "token_masking":
        md["ap"]["eval_mask"] = None
        md["behavior"]["eval_mask"] = None
    else:
        raise Exception("Training objective not implemented yet.")      # trainer/base.py:100-101
    return md


def synth_batch_mods(B, T, mods, seed, pad=None):
    """Synthetic batch for an arbitrary modality list: `mods` = [(name, channels)];
    'ap' draws Poisson(0.3) counts, every other modality N(0,1).  Same host generator recipe as synth_batch."""
    g = torch.Generator().manual_seed(seed)
    data = {}
    for name, n in mods:
        data[name] = torch.poisson(torch.full((B, T, n), 0.3), generator=g) if name == "ap" else torch.randn(B, T, n, generator=g)
    attn = torch.ones(B, T, dtype=torch.int64)
    if pad is not None:
        for b, pb in enumerate(pad):
            if pb:
                attn[b, T - pb:] = 0
    ts = torch.arange(T, dtype=torch.int64)[None].repeat(B, 1)
    return dict(data=data, time_attn_mask=attn, spikes_timestamps=ts)


def make_mod_dict_mods(batch, mods, masked=None):
    """mod_dict for an arbitrary modality list, built the way trainer/base.py:51-103 builds it for two.
    masked = None -> eval_mask None everywhere (token_masking: the model's masker draws the masks);
    masked = name  -> that modality fully masked (ones), all others zeros (encoding/decoding style)."""
    md = {}
    for i, (name, n) in enumerate(mods):
        x = batch["data"][name]
        md[name] = dict(inputs_modality=torch.tensor(i), inputs_attn_mask=batch["time_attn_mask"],
                        inputs_timestamp=batch["spikes_timestamps"], masking_mode=None,
                        inputs=x.clone(), targets=x.clone())
        if name == "ap":
            md[name]["inputs_regions"] = np.full((x.shape[0], n), "XX")
        if masked is None:
            md[name]["eval_mask"] = None
        else:
            md[name]["eval_mask"] = (torch.ones_like(x) if name == masked else torch.zeros_like(x)).to(torch.int64)
    return md


class OracleTrainer:
    """The four hot lines of train_epoch (trainer/base.py:191-198) around `forward`."""

    def __init__(self, sd, cfg: OracleCfg, masker_cfg: dict, total_steps: int, lr=1e-4, wd=0.01, eps=1e-8):
        self.cfg, self.sd = cfg, sd
        self.keys = trainable_keys(sd, cfg)
        for k in self.keys:
            sd[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(sd[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.keys}
        self.masker = OracleMasker(masker_cfg)
        self.total, self.lr, self.wd, self.eps, self.t = total_steps, lr, wd, eps, 0

    def step(self, batch, objective, training=True):
        out = forward(self.sd, make_mod_dict(batch, objective, self.cfg.avail_mod), self.cfg, training,
                      self.masker)
        grads = torch.autograd.grad(out["loss"], [self.sd[k] for k in self.keys])
        lr, b1 = onecycle(self.t, self.total, max_lr=self.lr)
        self.t += 1
        with torch.no_grad():
            for k, g in zip(self.keys, grads):
                adamw_step(self.sd[k], g, self.m[k], self.v[k], self.t, lr, b1, eps=self.eps, wd=self.wd)
        return out["loss"].detach()


def objective_schedule(n, seed=42):
    """`random.sample(schemes, 1)[0]` per step (trainer/base.py:189-190) after random.seed."""
    random.seed(seed)
    return [random.sample(["encoding", "decoding", "token_masking"], 1)[0] for _ in range(n)]

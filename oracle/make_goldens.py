#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
exists), never on the GPU box.  It imports the reference's own Python model
(`/root/reference/src/multi_modal/*`, `models/masker.py`, `trainer/base.py`),
feeds it seeded synthetic inputs and stores inputs + outputs as small .npz/.json
fixtures.  No reference source text is stored: fixtures are data only.

    cd /root/repo && python oracle/make_goldens.py

Fixture list follows SURVEY.md §8c.
"""
import json
import os
import random
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MMFM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
os.chdir(REF)  # the reference's config paths are cwd-relative

import numpy as np
import torch

torch.set_num_threads(8)

from multi_modal.mm import MultiModal  # noqa: E402  (reference)
from multi_modal.encoder_embeddings import EncoderEmbedding  # noqa: E402
from multi_modal.decoder_embeddings import DecoderEmbedding  # noqa: E402
from models.masker import Masker  # noqa: E402
from utils.config_utils import config_from_kwargs, update_config, DictConfig  # noqa: E402


# --------------------------------------------------------------------------- helpers
def ref_config():
    cfg = config_from_kwargs({"model": "include:src/configs/multi_modal/mm.yaml"})
    cfg = update_config("src/configs/multi_modal/trainer_mm.yaml", cfg)
    return cfg


def plain(d):
    """DictConfig -> plain nested dict (deep copy)."""
    return json.loads(json.dumps(d))


def tiny_model_cfg(H=32, heads=4, inter=64, n_enc=1, n_dec=1, max_F=8, dropout=0.0,
                   emb_dropout=0.0, sep=False, causal=False, n_modality=2):
    m = plain(ref_config()["model"])
    for side in ("encoder", "decoder"):
        m[side]["embedder"].update(max_F=max_F, dropout=emb_dropout, n_modality=n_modality)
        m[side]["transformer"].update(hidden_size=H, n_heads=heads, inter_size=inter,
                                      dropout=dropout)
    m["encoder"]["transformer"]["n_layers"] = n_enc
    m["decoder"]["transformer"]["n_layers"] = n_dec
    m["decoder"]["decoder_sep_mask"] = sep
    m["decoder"]["decoder_causal_mask"] = causal
    return DictConfig(m)


def build_model(model_cfg, n_ap, n_beh, seed):
    """Construction order of train_multi_modal.py:160-189."""
    torch.manual_seed(seed)
    enc, dec = {}, {}
    for mod in ("ap", "behavior"):
        enc[mod] = EncoderEmbedding(hidden_size=model_cfg.encoder.transformer.hidden_size,
                                    n_channel=n_ap if mod == "ap" else n_beh,
                                    config=model_cfg.encoder)
    for mod in ("ap", "behavior"):
        dec[mod] = DecoderEmbedding(hidden_size=model_cfg.decoder.transformer.hidden_size,
                                    n_channel=n_ap if mod == "ap" else n_beh,
                                    output_channel=n_ap if mod == "ap" else n_beh,
                                    config=model_cfg.decoder)
    return MultiModal(enc, dec, avail_mod=["ap", "behavior"], config=model_cfg,
                      share_modality_embeddings=True)


def synth_batch(B, T, n_ap, n_beh, seed, pad=None, ts_shift=None):
    """SURVEY.md §8d synthetic recipe; `pad[b]` = number of right-padded bins."""
    g = torch.Generator().manual_seed(seed)
    spikes = torch.poisson(torch.full((B, T, n_ap), 0.3), generator=g)
    beh = torch.randn(B, T, n_beh, generator=g)
    attn = torch.ones(B, T, dtype=torch.int64)
    if pad is not None:
        for b, p in enumerate(pad):
            if p:
                attn[b, T - p:] = 0
    ts = torch.arange(T, dtype=torch.int64)[None].repeat(B, 1)
    if ts_shift is not None:
        ts = (ts + torch.tensor(ts_shift)[:, None]) % T
    return dict(spikes_data=spikes, target=beh, time_attn_mask=attn, spikes_timestamps=ts)


def make_mod_dict(batch, objective, regions=None):
    """What MultiModalTrainer._forward_model_outputs builds (trainer/base.py:51-103)."""
    spikes, beh = batch["spikes_data"], batch["target"]
    B, T, n_ap = spikes.shape
    md = {}
    for i, mod in enumerate(("ap", "behavior")):
        d = dict(inputs_modality=torch.tensor(i), targets_modality=torch.tensor(i),
                 inputs_attn_mask=batch["time_attn_mask"],
                 inputs_timestamp=batch["spikes_timestamps"],
                 targets_timestamp=batch["spikes_timestamps"],
                 eid="synthetic", num_neuron=n_ap, masking_mode=None)
        x = spikes if mod == "ap" else beh
        d["inputs"], d["targets"] = x.clone(), x.clone()
        if mod == "ap":
            d["inputs_regions"] = regions if regions is not None else np.full((B, n_ap), "XX")
        md[mod] = d
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
        raise ValueError(objective)
    return md


def npify(t):
    return t.detach().cpu().numpy()


def save_npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def save_json(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- fixtures
def fx_init_order():
    cfg = ref_config()
    model = build_model(cfg.model, 668, 2, seed=42)
    entries = []
    for k, v in model.state_dict().items():
        f = v.detach().double().flatten()
        entries.append(dict(key=k, shape=list(v.shape), sum=float(f.sum()),
                            abssum=float(f.abs().sum()),
                            first=[float(x) for x in v.detach().flatten()[:4]]))
    named = [k for k, _ in model.named_parameters()]
    save_json("init_order.json", dict(
        seed=42, n_state=len(entries), n_param=len(named),
        numel_state=int(sum(np.prod(e["shape"]) for e in entries)),
        numel_param=int(sum(p.numel() for p in model.parameters())),
        named_parameters=named, state_dict=entries))


def fx_tiny_fwd_bwd():
    """Tiny config: every tensor of the forward, loss and all grads, three objectives."""
    B, T, n_ap, n_beh = 2, 8, 12, 2
    arrs = {}
    meta = dict(B=B, T=T, n_ap=n_ap, n_beh=n_beh, H=32, heads=4, inter=64,
                n_enc=1, n_dec=1, max_F=8, model_seed=7, data_seed=3, cases=[])
    variants = [
        ("base", dict(), None, None),
        ("pad", dict(), [0, 2], [0, 3]),          # sample 1 has 2 padded bins, shifted stamps
        ("sep", dict(sep=True), [0, 2], None),
        ("causal", dict(causal=True), [0, 2], None),
        ("deep", dict(n_enc=2, n_dec=2), None, None),
    ]
    for vname, kw, pad, shift in variants:
        mcfg = tiny_model_cfg(**kw)
        model = build_model(mcfg, n_ap, n_beh, seed=7)
        model.train()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k, v in sd.items():
            arrs[f"{vname}/sd/{k}"] = npify(v)
        batch = synth_batch(B, T, n_ap, n_beh, seed=3, pad=pad, ts_shift=shift)
        for k, v in batch.items():
            arrs[f"{vname}/batch/{k}"] = npify(v)
        for obj in ("encoding", "decoding", "token_masking"):
            caps = {}
            hooks = []

            def cap(name):
                def fn(_m, _i, o):
                    caps[name] = o.detach().clone()
                return fn
            def cap2(name):
                def fn(_m, _i, o):
                    caps[name + "_x"], caps[name + "_emb"] = o[0].detach().clone(), o[1].detach().clone()
                return fn
            for mod in ("ap", "behavior"):
                hooks.append(model.encoder_embeddings[mod].embedder.register_forward_hook(cap2(f"enc_embed/{mod}")))
                hooks.append(model.decoder_embeddings[mod].embedder.register_forward_hook(cap2(f"dec_embed/{mod}")))
            hooks.append(model.encoder_norm.register_forward_hook(cap("enc_out")))
            hooks.append(model.decoder_proj_context.register_forward_hook(cap("ctx_proj")))
            hooks.append(model.decoder_norm.register_forward_hook(cap("dec_out")))
            model.zero_grad(set_to_none=True)
            torch.manual_seed(11)      # masker stream for token_masking
            md = make_mod_dict(batch, obj)
            out = model(md)
            out.loss.backward()
            for h in hooks:
                h.remove()
            p = f"{vname}/{obj}"
            arrs[f"{p}/loss"] = npify(out.loss)
            for mod in ("ap", "behavior"):
                arrs[f"{p}/mod_loss/{mod}"] = npify(out.mod_loss[mod])
                arrs[f"{p}/n/{mod}"] = npify(out.mod_n_examples[mod])
                arrs[f"{p}/preds/{mod}"] = npify(out.mod_preds[mod])
                arrs[f"{p}/mask/{mod}"] = npify(md[mod]["inputs_mask"])
            for k, v in caps.items():
                arrs[f"{p}/{k}"] = npify(v)
            for k, prm in model.named_parameters():
                arrs[f"{p}/grad/{k}"] = npify(prm.grad)
            meta["cases"].append(p)
    arrs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save_npz("tiny_fwd_bwd.npz", **arrs)


def default_batch():
    return synth_batch(16, 100, 668, 2, seed=0)


def fx_default_scalars():
    cfg = ref_config()
    model = build_model(cfg.model, 668, 2, seed=42)
    model.eval()
    batch = default_batch()
    res = {}
    for obj in ("encoding", "decoding", "token_masking"):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(1)
        out = model(make_mod_dict(batch, obj))
        out.loss.backward()
        res[obj] = dict(
            loss=float(out.loss),
            mod_loss={m: float(v) for m, v in out.mod_loss.items()},
            n={m: int(v) for m, v in out.mod_n_examples.items()},
            pred_abssum={m: float(v.double().abs().sum()) for m, v in out.mod_preds.items()},
            grad_norm={k: float(p.grad.double().norm()) for k, p in model.named_parameters()})
        print("   ", obj, res[obj]["loss"], res[obj]["n"])
    save_json("default_scalars.json", res)


def fx_masker_bits():
    base = plain(ref_config()["model"]["masker"])
    arrs, cases = {}, []
    cid = 0
    for seed in (0, 5):
        for ratio in (0.1, 0.3):
            for expand_prob, max_ts in ((0.0, 1), (1.0, 3)):
                mc = dict(base, ratio=ratio, expand_prob=expand_prob, max_timespan=max_ts)
                mk = Masker(DictConfig(mc))
                mk.train()
                torch.manual_seed(seed)
                g = torch.Generator().manual_seed(100 + seed)
                ap = torch.poisson(torch.full((4, 20, 9), 0.3), generator=g)
                bh = torch.randn(4, 20, 2, generator=g)
                regions = np.full((4, 9), "XX")
                _, m_ap = mk(ap.clone(), regions)          # first call (ap)
                _, m_bh = mk(bh.clone(), None)             # second call (behaviour)
                after = torch.rand(3)                      # pins the generator state afterwards
                arrs[f"c{cid}/ap"] = npify(ap)
                arrs[f"c{cid}/bh"] = npify(bh)
                arrs[f"c{cid}/mask_ap"] = npify(m_ap)
                arrs[f"c{cid}/mask_bh"] = npify(m_bh)
                arrs[f"c{cid}/after"] = npify(after)
                cases.append(dict(id=cid, seed=seed, cfg=mc))
                cid += 1
    # early-outs (masker.py:62-69)
    mk = Masker(DictConfig(dict(base, ratio=0)))
    x = torch.ones(2, 3, 4)
    _, z = mk(x.clone(), None)
    arrs["zero_ratio_mask"] = npify(z)
    arrs["meta"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save_npz("masker_bits.npz", **arrs)


def fx_mask_index_ops():
    """forward_mask_encoder / forward_mask_decoder on hand-made masks (mm.py:141-194)."""
    B, T, H = 3, 6, 4
    arrs, cases = {}, []
    g = torch.Generator().manual_seed(9)
    hand = {
        "none": torch.zeros(B, T, dtype=torch.int64),
        "one": torch.zeros(B, T, dtype=torch.int64),
        "many": (torch.rand(B, T, generator=g) < 0.5).to(torch.int64),
        "all": torch.ones(B, T, dtype=torch.int64),
    }
    hand["one"][0, 2] = 1
    hand["one"][1, 4] = 1
    attn = torch.ones(B, T, dtype=torch.int64)
    attn[1, 4:] = 0
    attn[2, 5:] = 0
    for sep in (False, True):
        for causal in (False, True):
            model = build_model(tiny_model_cfg(H=H, heads=2, inter=8, max_F=T, sep=sep, causal=causal),
                                5, 2, seed=1)
            for name, mk in hand.items():
                md = {}
                for i, mod in enumerate(("ap", "behavior")):
                    msk = (mk if mod == "ap" else mk.flip(1)) & attn
                    md[mod] = dict(x=torch.randn(B, T, H, generator=g), emb=torch.randn(B, T, H, generator=g),
                                   gt=torch.zeros(B, T, 1), inputs_mask=msk, targets_mask=msk,
                                   encoder_attn_mask=attn, decoder_attn_mask=attn)
                key = f"sep{int(sep)}_causal{int(causal)}/{name}"
                for mod in md:
                    arrs[f"{key}/in_x/{mod}"] = npify(md[mod]["x"])
                    arrs[f"{key}/in_mask/{mod}"] = npify(md[mod]["inputs_mask"])
                et, ee, em, eam, emm = model.forward_mask_encoder(md)
                dt, _, de, dm, dam, dmm = model.forward_mask_decoder(md)
                arrs[f"{key}/enc_tokens"] = npify(et)
                arrs[f"{key}/enc_mask"] = npify(em)
                arrs[f"{key}/enc_attn_mask"] = npify(eam)
                arrs[f"{key}/enc_mod_mask"] = npify(emm)
                arrs[f"{key}/dec_tokens"] = npify(dt)
                arrs[f"{key}/dec_attn_mask"] = npify(dam.to(torch.int64))
                arrs[f"{key}/dec_mod_mask"] = npify(dmm)
                cases.append(key)
    arrs["attn"] = npify(attn)
    arrs["meta"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save_npz("mask_index_ops.npz", **arrs)


def make_opt(model, total_steps, lr=1e-4, wd=0.01, eps=1e-8):
    from torch.optim.lr_scheduler import OneCycleLR
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd, eps=eps)
    sch = OneCycleLR(optimizer=opt, total_steps=total_steps, max_lr=lr, pct_start=0.15, div_factor=10)
    return opt, sch


def fx_sched_adamw():
    lin = torch.nn.Linear(2, 2)
    opt, sch = make_opt(lin, 1000)
    lrs, b1s = [], []
    for _ in range(1000):
        lrs.append(opt.param_groups[0]["lr"])
        b1s.append(opt.param_groups[0]["betas"][0])
        opt.step()
        sch.step()
    arrs = dict(lr=np.asarray(lrs, dtype=np.float64), beta1=np.asarray(b1s, dtype=np.float64))
    # 5-step AdamW trajectory on the tiny model, encoding objective, total_steps=20
    B, T, n_ap, n_beh = 2, 8, 12, 2
    model = build_model(tiny_model_cfg(), n_ap, n_beh, seed=7)
    model.train()
    opt, sch = make_opt(model, 20)
    for step in range(5):
        batch = synth_batch(B, T, n_ap, n_beh, seed=step)
        out = model(make_mod_dict(batch, "encoding"))
        out.loss.backward()
        opt.step()
        sch.step()
        opt.zero_grad()
        arrs[f"traj/loss{step}"] = npify(out.loss)
    for k, v in model.state_dict().items():
        arrs[f"traj/final/{k}"] = npify(v)
    save_npz("sched_adamw.npz", **arrs)


def run_curve(model, steps, B, T, n_ap, n_beh, total_steps):
    opt, sch = make_opt(model, total_steps)
    model.train()
    random.seed(42)
    torch.manual_seed(1234)
    losses, objs = [], []
    for step in range(steps):
        obj = random.sample(["encoding", "decoding", "token_masking"], 1)[0]
        batch = synth_batch(B, T, n_ap, n_beh, seed=step)
        out = model(make_mod_dict(batch, obj))
        out.loss.backward()
        opt.step()
        sch.step()
        opt.zero_grad()
        losses.append(float(out.loss))
        objs.append(obj)
    return losses, objs


def fx_loss_curve():
    res = {}
    model = build_model(tiny_model_cfg(), 12, 2, seed=7)
    l, o = run_curve(model, 50, 2, 8, 12, 2, total_steps=50)
    res["tiny"] = dict(loss=l, objective=o, model_seed=7, B=2, T=8, n_ap=12, n_beh=2, total_steps=50)
    cfg = plain(ref_config()["model"])
    for side in ("encoder", "decoder"):
        cfg[side]["embedder"]["dropout"] = 0.0
        cfg[side]["transformer"]["dropout"] = 0.0
    model = build_model(DictConfig(cfg), 668, 2, seed=42)
    l, o = run_curve(model, 30, 16, 100, 668, 2, total_steps=1000)
    res["default"] = dict(loss=l, objective=o, model_seed=42, B=16, T=100, n_ap=668, n_beh=2,
                          total_steps=1000)
    print("    default curve:", l[:3], "...", l[-1])
    save_json("loss_curve.json", res)


def fx_loss_curve_1k():
    """North-star curve length: 1000 optimisation steps of the reference (tiny model, dropout 0, mixed objectives)."""
    model = build_model(tiny_model_cfg(), 12, 2, seed=7)
    l, o = run_curve(model, 1000, 2, 8, 12, 2, total_steps=1000)
    print("    tiny 1k curve:", l[:2], "...", l[-2:])
    save_json("loss_curve_1k.json", dict(loss=l, objective=o, model_seed=7, B=2, T=8, n_ap=12, n_beh=2, total_steps=1000))


CONFIG5_MODS = [("ap", 668), ("behavior", 2), ("lfp", 128)]


def fx_config5_scalars():
    """BASELINE.json configs[4] as SURVEY.md §8d instantiates it: H=512, I=1024, 8 heads (dh=64), T=200, three
    modalities (L=600), 5+5 layers.  The reference defines losses for 'ap'/'behavior' only (mm.py:79-82), so 'lfp'
    gets nn.MSELoss(reduction='none') added to model.loss_mod, as §8d prescribes.  B=2, eval mode, one padded trial."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mm_oracle as O
    m = plain(ref_config()["model"])
    for side in ("encoder", "decoder"):
        m[side]["embedder"].update(max_F=200, n_modality=3)
        m[side]["transformer"].update(hidden_size=512, n_heads=8, inter_size=1024)
    cfg = DictConfig(m)
    torch.manual_seed(42)
    enc = {mod: EncoderEmbedding(hidden_size=512, n_channel=n, config=cfg.encoder) for mod, n in CONFIG5_MODS}
    dec = {mod: DecoderEmbedding(hidden_size=512, n_channel=n, output_channel=n, config=cfg.decoder) for mod, n in CONFIG5_MODS}
    model = MultiModal(enc, dec, avail_mod=[mod for mod, _ in CONFIG5_MODS], config=cfg, share_modality_embeddings=True)
    model.loss_mod["lfp"] = torch.nn.MSELoss(reduction="none")
    model.eval()
    batch = O.synth_batch_mods(2, 200, CONFIG5_MODS, seed=0, pad=[0, 7])
    res = dict(mods=CONFIG5_MODS, B=2, T=200, pad=[0, 7], model_seed=42, batch_seed=0, masker_seed=1, cases={})
    for case, masked in (("token_masking", None), ("mask_ap", "ap"), ("mask_lfp", "lfp")):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(1)
        out = model(O.make_mod_dict_mods(batch, CONFIG5_MODS, masked))
        out.loss.backward()
        res["cases"][case] = dict(
            loss=float(out.loss),
            mod_loss={k: float(v) for k, v in out.mod_loss.items()},
            n={k: int(v) for k, v in out.mod_n_examples.items()},
            pred_abssum={k: float(v.double().abs().sum()) for k, v in out.mod_preds.items()},
            grad_norm={k: float(p.grad.double().norm()) for k, p in model.named_parameters()})
        print("   ", case, res["cases"][case]["loss"], res["cases"][case]["n"])
    # a few raw values so a permutation of tokens/modalities cannot hide behind the sums
    res["pred_samples"] = {k: [float(x) for x in out.mod_preds[k][1, 150:153, 0]] for k in out.mod_preds}
    save_json("config5_scalars.json", res)


def _stub_wandb_torcheval():
    """wandb and torcheval are not installed here: utils/metric_utils.py and trainer/base.py import them at module load.  The
    stand-ins are shared by every fixture that imports those modules (first registration wins in sys.modules); R2Score is a
    working 1 - SS_res / SS_tot so that the trainer fixture can run its evaluation."""
    wb = types.ModuleType("wandb")
    wb.log = lambda *a, **k: None
    wb.Image = lambda x: x
    sys.modules.setdefault("wandb", wb)
    te = types.ModuleType("torcheval")
    tem = types.ModuleType("torcheval.metrics")

    class R2Score:
        def reset(self):
            self.p, self.t = [], []

        def to(self, d):
            return self

        def update(self, p, t):
            self.p.append(p)
            self.t.append(t)

        def compute(self):
            p, t = torch.cat(self.p), torch.cat(self.t)
            return 1 - ((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum()
    tem.R2Score = R2Score
    te.metrics = tem
    sys.modules.setdefault("torcheval", te)
    sys.modules.setdefault("torcheval.metrics", tem)


def fx_trainer_io():
    """The reference trainer on 3 synthetic batches (wandb / torcheval stubbed)."""
    import transformers  # noqa: F401  (import before stubbing)
    _stub_wandb_torcheval()
    import matplotlib
    matplotlib.use("Agg")
    from trainer.make import make_multimodal_trainer

    B, T, n_ap, n_beh = 4, 8, 12, 2
    cfg = ref_config()
    model = build_model(tiny_model_cfg(), n_ap, n_beh, seed=7)

    def loader(seed0):
        out = []
        for i in range(3):
            b = synth_batch(B, T, n_ap, n_beh, seed=seed0 + i)
            b["eid"] = ["synthetic"] * B
            b["neuron_regions"] = [["XX"] * B for _ in range(n_ap)]
            out.append(b)
        return out
    opt, sch = make_opt(model, 100)

    class Acc:
        device = torch.device("cpu")
    tr = make_multimodal_trainer(model=model, train_dataloader=loader(0), eval_dataloader=loader(50),
                                 optimizer=opt, log_dir="/tmp", accelerator=Acc(), lr_scheduler=sch,
                                 avail_mod=["ap", "behavior"],
                                 modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]),
                                 mixed_training=True, config=cfg, num_neurons=[n_ap])
    random.seed(42)
    torch.manual_seed(99)
    st = random.getstate()
    objs = [random.sample(["encoding", "decoding", "token_masking"], 1)[0] for _ in range(6)]
    random.setstate(st)
    tr_res = tr.train_epoch(0)
    ev = tr.eval_epoch()
    save_json("trainer_io.json", dict(
        B=B, T=T, n_ap=n_ap, n_beh=n_beh, model_seed=7, objectives=objs,
        train_loss=float(tr_res["train_loss"]), eval_loss=float(ev["eval_loss"]),
        eval_keys=sorted(ev.keys()),
        eval_gt_shapes={m: list(ev["eval_gt"][0][m].shape) for m in ev["eval_gt"][0]},
        eval_preds_shapes={m: list(ev["eval_preds"][0][m].shape) for m in ev["eval_preds"][0]},
        eval_preds_abssum={m: float(ev["eval_preds"][0][m].double().abs().sum()) for m in ev["eval_preds"][0]},
        eval_trial_avg_r2=float(ev["eval_trial_avg_r2"])))


def fx_loader_collate():
    """The reference's BaseDataset.__getitem__ (loader/base.py:304-450) on ragged synthetic CSR trials."""
    from scipy.sparse import csr_array
    import datasets
    if not hasattr(datasets, "list_datasets"):      # removed from the installed `datasets` release; the reference's
        datasets.list_datasets = lambda *a, **k: []  # utils/dataset_utils.py imports the name at module load (unused here)
    from loader.base import BaseDataset
    rng = np.random.default_rng(5)
    max_T, max_N, pad = 10, 14, -1.0
    trials, arrs, meta = [], {}, []
    for i, (T_i, N_i) in enumerate([(10, 14), (7, 14), (10, 9), (6, 5), (13, 14), (10, 17), (12, 20), (2, 3)]):
        dense = (rng.random((T_i, N_i)) < 0.25) * rng.integers(1, 4, (T_i, N_i))
        sp = csr_array(dense.astype(np.uint8))
        d = dict(spikes_sparse_data=sp.data.tolist(), spikes_sparse_indices=sp.indices.tolist(),
                 spikes_sparse_indptr=sp.indptr.tolist(), spikes_sparse_shape=list(sp.shape),
                 choice=float(i % 2), block=0.2 * i, reward=float((i + 1) % 2), eid=f"eid{i}",
                 cluster_depths=rng.random(N_i).tolist(), cluster_regions=[f"R{j % 3}" for j in range(N_i)])
        d["wheel-speed"] = rng.standard_normal(T_i).tolist()
        d["whisker-motion-energy"] = rng.standard_normal(T_i).tolist()
        trials.append(d)
    ds = BaseDataset(dataset=trials, target=["wheel-speed", "whisker-motion-energy"], pad_value=pad, max_time_length=max_T,
                     max_space_length=max_N, pad_to_right=True, load_meta=True, dataset_name="ibl")
    for i, d in enumerate(trials):
        out = ds[i]
        for k in ("spikes_sparse_data", "spikes_sparse_indices", "spikes_sparse_indptr", "spikes_sparse_shape", "wheel-speed",
                  "whisker-motion-energy", "cluster_depths"):
            arrs[f"t{i}/in/{k}"] = np.asarray(d[k])
        for k in ("spikes_data", "time_attn_mask", "space_attn_mask", "spikes_timestamps", "spikes_spacestamps", "target",
                  "neuron_depths", "choice", "block", "reward"):
            arrs[f"t{i}/out/{k}"] = np.asarray(out[k])
        meta.append(dict(eid=out["eid"], regions_in=d["cluster_regions"], regions_out=[str(x) for x in out["neuron_regions"]]))
    arrs["meta"] = np.frombuffer(json.dumps(dict(max_T=max_T, max_N=max_N, pad=pad, trials=meta)).encode(), dtype=np.uint8)
    save_npz("loader_collate.npz", **arrs)


MULTISESSION = dict(neurons=[14, 9, 11, 14, 6, 12], trials=3, T=8, max_N=14, pad=-1.0, epochs=2, model_seed=7)


def fx_multisession_curve():
    """BASELINE configs[2]: multi-session pretraining = single-session batches (trainer/base.py:65) of sessions with
    different neuron counts, right-padded by the loader to max_space_length with pad_value=-1 (train_multi_modal.py:
    121-128).  The reference's make_loader -> BaseDataset -> default_collate feeds the reference model; 12 steps."""
    import datasets
    if not hasattr(datasets, "list_datasets"):
        datasets.list_datasets = lambda *a, **k: []
    from loader.make_loader import make_loader
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import loader_oracle as LO
    ms = MULTISESSION
    loaders = []
    for s_id, n in enumerate(ms["neurons"]):
        trials = LO.synth_session_trials(n, ms["trials"], ms["T"], seed=100 + s_id, eid=f"session{s_id}")
        loaders.append(make_loader(trials, batch_size=ms["trials"], target=["wheel-speed", "whisker-motion-energy"], pad_value=ms["pad"],
                                   max_time_length=ms["T"], max_space_length=ms["max_N"], load_meta=True, shuffle=False))
    model = build_model(tiny_model_cfg(), ms["max_N"], 2, seed=ms["model_seed"])
    total = ms["epochs"] * len(loaders)
    opt, sch = make_opt(model, total)
    model.train()
    batches = [next(iter(l)) for l in loaders]        # (creating a DataLoader iterator draws from the default generator)
    random.seed(42)
    torch.manual_seed(1234)
    losses, objs, ns = [], [], []
    for step in range(total):
        batch = batches[step % len(loaders)]
        obj = random.sample(["encoding", "decoding", "token_masking"], 1)[0]
        regions = np.asarray(batch["neuron_regions"]).T                 # trainer/base.py:57
        b = dict(spikes_data=batch["spikes_data"].float(), target=batch["target"].float(), time_attn_mask=batch["time_attn_mask"],
                 spikes_timestamps=batch["spikes_timestamps"])
        out = model(make_mod_dict(b, obj, regions=regions))
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(float(out.loss.detach()))
        objs.append(obj)
        ns.append({k: int(v) for k, v in out.mod_n_examples.items()})
    print("    multisession curve:", losses[:3], "...", losses[-1])
    save_json("multisession_curve.json", dict(ms, loss=losses, objective=objs, n=ns,
                                              final_norm={k: float(v.double().norm()) for k, v in model.state_dict().items()}))


def fx_eval_metrics():
    """The reference's own bits_per_spike / neg_log_likelihood (utils/eval_utils.py:1051-1119) on seeded rate / spike arrays.
    eval_utils imports torcheval (absent) through utils.metric_utils at module load: the shared stand-in satisfies the import;
    nothing of it runs in these functions."""
    import datasets
    if not hasattr(datasets, "list_datasets"):
        datasets.list_datasets = lambda *a, **k: []
    _stub_wandb_torcheval()
    import matplotlib
    matplotlib.use("Agg")
    from utils.eval_utils import bits_per_spike, neg_log_likelihood
    rng = np.random.default_rng(11)
    arrs, cases = {}, []
    for i, (B, T, N, scale) in enumerate([(3, 5, 4, 0.3), (8, 100, 50, 0.3), (4, 20, 7, 2.0)]):
        spikes = rng.poisson(scale, (B, T, N)).astype(np.float64)
        rates = np.exp(rng.standard_normal((B, T, N)) * 0.5 + np.log(scale))
        if i == 2:
            rates[0, 0, 0] = 0.0                      # the zero-rate replacement path (1e-9)
            spikes[:, :, 3] = 0.0                     # a silent neuron: null rate 0 -> 1e-9
        arrs[f"c{i}/rates"], arrs[f"c{i}/spikes"] = rates.astype(np.float32), spikes.astype(np.float32)
        r32, s32 = arrs[f"c{i}/rates"].astype(np.float64), arrs[f"c{i}/spikes"].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):      # the per-neuron loop of spiking_activity_recon_eval (eval_utils.py:846-851)
            per = [float(bits_per_spike(r32[:, :, [n]].copy(), s32[:, :, [n]])) for n in range(N)]
        arrs[f"c{i}/bps_per_neuron"] = np.asarray([np.nan if np.isinf(b) else b for b in per], dtype=np.float64)
        cases.append(dict(id=i, bps=float(bits_per_spike(r32.copy(), s32)), nll=float(neg_log_likelihood(r32.copy(), s32))))
        print("   ", cases[-1])
    # heldout_mask (utils/eval_utils.py:988-1045), every mode, on a small seeded array
    from utils.eval_utils import heldout_mask
    sp = torch.from_numpy(rng.poisson(0.8, (3, 6, 10)).astype(np.float32))
    regions = np.array(["CA1", "PO", "CA1", "LP", "PO", "CA1", "LP", "PO", "CA1", "DG"])
    hm_cases = [dict(mode="manual", heldout_idxs=[1, 4, 7]), dict(mode="most", n_active=3),
                dict(mode="inter_region", heldout_idxs=[0, 2], target_regions=["CA1", "PO"]),
                dict(mode="intra_region", heldout_idxs=[1], target_regions=["CA1"]),
                dict(mode="intra_region", heldout_idxs=[], target_regions=["LP"]),
                dict(mode="forward_pred", heldout_idxs=[4, 5]), dict(mode="modal_spike", heldout_idxs=[0, 3])]
    arrs["hm/spikes"], arrs["hm/regions"] = sp.numpy(), np.frombuffer(json.dumps(regions.tolist()).encode(), dtype=np.uint8)
    for j, kw in enumerate(hm_cases):
        call = {k: (np.array(v) if k == "heldout_idxs" else v) for k, v in kw.items()}
        out = heldout_mask(sp.clone(), neuron_regions=regions, **call)
        arrs[f"hm/{j}/spikes"], arrs[f"hm/{j}/eval_mask"] = out["spikes"].numpy(), out["eval_mask"].numpy()
        arrs[f"hm/{j}/heldout_idxs"] = np.asarray(out["heldout_idxs"], dtype=np.int64)
    arrs["meta"] = np.frombuffer(json.dumps(dict(cases=cases, heldout=hm_cases)).encode(), dtype=np.uint8)
    save_npz("eval_metrics.npz", **arrs)


def fx_masker_modes():
    """Every Masker mode (models/masker.py:78-168) beyond `temporal`: returned spikes AND target masks, plus the generator
    state afterwards (torch CPU generator and Python's `random`, which the region modes draw from)."""
    base = plain(ref_config()["model"]["masker"])
    regions_row = ["CA1", "PO", "CA1", "LP", "PO", "CA1", "LP", "PO", "DG"]
    cases_in = [
        dict(mode="neuron", ratio=0.3),
        dict(mode="random", ratio=0.2),
        dict(mode="co-smooth", ratio=0.3, channels=[1, 4, 7]),
        dict(mode="forward-pred", ratio=0.3, timesteps=[15, 16, 17, 18, 19]),
        dict(mode="inter-region", ratio=0.3, mask_regions=["CA1", "PO", "LP"], target_regions=["all"], n_mask_regions=2),
        dict(mode="intra-region", ratio=0.4, mask_regions=["all"], target_regions=["CA1", "PO"], n_mask_regions=1),
        dict(mode="causal", ratio=0.3, max_timespan=3, causal_zero=True),
        dict(mode="causal", ratio=0.3, max_timespan=2, causal_zero=False),
        dict(mode="temporal", ratio=0.3, expand_prob=1.0, max_timespan=4, zero_ratio=0.7, random_ratio=0.5),
        dict(mode="neuron", ratio=0.5, zero_ratio=0.5, random_ratio=1.0),
    ]
    arrs, cases = {}, []
    for cid, kw in enumerate(cases_in):
        mc = dict(base, **kw)
        mk = Masker(DictConfig(mc))
        mk.train()
        torch.manual_seed(3 + cid)
        random.seed(17 + cid)
        g = torch.Generator().manual_seed(200 + cid)
        ap = torch.poisson(torch.full((4, 20, 9), 0.6), generator=g)
        regions = np.asarray([regions_row] * 4)
        out1, m1 = mk(ap.clone(), regions)
        out2, m2 = mk(ap.clone(), regions)               # second call: generator / `random` consumption order
        arrs[f"c{cid}/ap"] = npify(ap)
        arrs[f"c{cid}/out1"], arrs[f"c{cid}/mask1"] = npify(out1), npify(m1)
        arrs[f"c{cid}/out2"], arrs[f"c{cid}/mask2"] = npify(out2), npify(m2)
        arrs[f"c{cid}/after"] = npify(torch.rand(3))
        cases.append(dict(id=cid, cfg=mc, after_random=random.random(), regions=regions_row))
    arrs["meta"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save_npz("masker_modes.npz", **arrs)


def fx_masker_modes_model():
    """f4 on the model path: every masker mode through the reference's MultiModal.forward / backward with eval_mask = None
    (the model draws its own masks, mm.py:262-267 keeps channel 0 of the masker's mask), the mode set by mutating
    `model.masker` AFTER construction as utils/eval_utils.py:63-67 does.  Stored: inputs, state dict, per-modality mask, exact n,
    loss, predictions and every gradient norm.  A mode the reference cannot run on this two-modality batch is stored as
    raises = <exception type name>."""
    B, T, n_ap, n_beh = 3, 20, 9, 2
    regions_row = ["CA1", "PO", "CA1", "LP", "PO", "CA1", "LP", "PO", "DG"]
    cases_in = [
        dict(mode="neuron", ratio=0.3),
        dict(mode="random", ratio=0.2),
        dict(mode="co-smooth", ratio=0.3, channels=[1]),
        dict(mode="co-smooth", ratio=0.3, channels=[1, 4, 7]),
        dict(mode="forward-pred", ratio=0.3, timesteps=[15, 16, 17, 18, 19]),
        dict(mode="inter-region", ratio=0.3, mask_regions=["CA1", "PO", "LP"], target_regions=["all"], n_mask_regions=2),
        dict(mode="intra-region", ratio=0.4, mask_regions=["all"], target_regions=["CA1", "PO"], n_mask_regions=1),
        dict(mode="causal", ratio=0.3, max_timespan=3, causal_zero=True),
        dict(mode="causal", ratio=0.3, max_timespan=2, causal_zero=False),
        dict(mode="temporal", ratio=0.3, expand_prob=1.0, max_timespan=4),
    ]
    arrs, cases = {}, []
    mcfg = tiny_model_cfg(max_F=T)
    batch = synth_batch(B, T, n_ap, n_beh, seed=5, pad=[0, 3, 0])
    for k, v in batch.items():
        arrs[f"batch/{k}"] = npify(v)
    regions = np.asarray([regions_row] * B)
    for cid, kw in enumerate(cases_in):
        model = build_model(mcfg, n_ap, n_beh, seed=9)
        if cid == 0:
            for k, v in model.state_dict().items():
                arrs[f"sd/{k}"] = npify(v)
        for k, v in kw.items():
            setattr(model.masker, k, v)
        model.train()
        torch.manual_seed(31 + cid)
        random.seed(41 + cid)
        md = make_mod_dict(batch, "token_masking", regions=regions)
        rec = dict(id=cid, set=kw)
        try:
            out = model(md)
            out.loss.backward()
        except Exception as e:  # noqa: BLE001 - the fixture records what upstream does
            rec["raises"] = type(e).__name__
            cases.append(rec)
            continue
        rec["raises"] = None
        rec["loss"] = float(out.loss)
        rec["n"] = {m: int(out.mod_n_examples[m]) for m in ("ap", "behavior")}
        rec["mod_loss"] = {m: float(out.mod_loss[m]) for m in ("ap", "behavior")}
        rec["grad_norm"] = {k: (0.0 if p.grad is None else float(p.grad.double().norm())) for k, p in model.named_parameters()}
        rec["after_rand"] = float(torch.rand(1))
        rec["after_random"] = random.random()
        for m in ("ap", "behavior"):
            arrs[f"c{cid}/mask/{m}"] = npify(md[m]["inputs_mask"])
            arrs[f"c{cid}/preds/{m}"] = npify(out.mod_preds[m])
        cases.append(rec)
    arrs["meta"] = np.frombuffer(json.dumps(dict(B=B, T=T, n_ap=n_ap, n_beh=n_beh, regions=regions_row, model_seed=9, cases=cases)).encode(),
                                 dtype=np.uint8)
    save_npz("masker_modes_model.npz", **arrs)
    print("   cases:", [(c["set"]["mode"], c["raises"]) for c in cases])


def fx_loss_curve_1k_default():
    """The north star's curve at the metric's own config: d_model 256, 5+5 layers, T=100, 668+2 channels, B=16, dropout 0,
    1000 optimisation steps of the reference on the CPU (mixed objectives, OneCycleLR over the 1000 steps)."""
    cfg = plain(ref_config()["model"])
    for side in ("encoder", "decoder"):
        cfg[side]["embedder"]["dropout"] = 0.0
        cfg[side]["transformer"]["dropout"] = 0.0
    model = build_model(DictConfig(cfg), 668, 2, seed=42)
    l, o = run_curve(model, 1000, 16, 100, 668, 2, total_steps=1000)
    print("    default 1k curve:", l[:2], "...", l[-2:])
    save_json("loss_curve_1k_default.json", dict(loss=l, objective=o, model_seed=42, B=16, T=100, n_ap=668, n_beh=2, total_steps=1000))


def fx_h64_curve():
    """BASELINE.json configs[0] as worded ("2-layer d_model=64"): MultiModal with 2+2 layers, hidden 64, 8 heads, inter 128, on the
    reference's CPU path: per-objective scalars (loss, n, every gradient norm) and a 100-step curve."""
    mc = tiny_model_cfg(H=64, heads=8, inter=128, n_enc=2, n_dec=2, max_F=100)
    model = build_model(mc, 668, 2, seed=11)
    batch = synth_batch(16, 100, 668, 2, seed=0)
    res = dict(model_seed=11, B=16, T=100, n_ap=668, n_beh=2, scalars={})
    model.eval()
    for obj in ("encoding", "decoding", "token_masking"):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(1)
        out = model(make_mod_dict(batch, obj))
        out.loss.backward()
        res["scalars"][obj] = dict(loss=float(out.loss), n={m: int(v) for m, v in out.mod_n_examples.items()},
                                   pred_abssum={m: float(v.double().abs().sum()) for m, v in out.mod_preds.items()},
                                   grad_norm={k: float(p.grad.double().norm()) for k, p in model.named_parameters()})
    model = build_model(mc, 668, 2, seed=11)
    l, o = run_curve(model, 100, 16, 100, 668, 2, total_steps=100)
    res.update(loss=l, objective=o, total_steps=100)
    print("    h64 curve:", l[:2], "...", l[-1])
    save_json("h64_curve.json", res)


MULTISESSION_BIG = dict(sessions=40, trials=4, T=100, max_N=668, lo=300, pad=-1.0, model_seed=42, neuron_seed=2024)


def fx_multisession_big():
    """BASELINE configs[2] at its stated size: 40 sessions of 300-668 neurons, single-session batches right-padded
    with -1 to 668 by the reference's own loader, default model (d_model 256, dropout 0), one pass over the sessions."""
    import datasets
    if not hasattr(datasets, "list_datasets"):
        datasets.list_datasets = lambda *a, **k: []
    from loader.make_loader import make_loader
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import loader_oracle as LO
    ms = MULTISESSION_BIG
    rng = np.random.default_rng(ms["neuron_seed"])
    neurons = [int(x) for x in rng.integers(ms["lo"], ms["max_N"] + 1, ms["sessions"])]
    cfg = plain(ref_config()["model"])
    for side in ("encoder", "decoder"):
        cfg[side]["embedder"]["dropout"] = 0.0
        cfg[side]["transformer"]["dropout"] = 0.0
    model = build_model(DictConfig(cfg), ms["max_N"], 2, seed=ms["model_seed"])
    opt, sch = make_opt(model, ms["sessions"])
    model.train()
    batches = []
    for s_id, n in enumerate(neurons):
        trials = LO.synth_session_trials(n, ms["trials"], ms["T"], seed=500 + s_id, eid=f"session{s_id}")
        ld = make_loader(trials, batch_size=ms["trials"], target=["wheel-speed", "whisker-motion-energy"], pad_value=ms["pad"],
                         max_time_length=ms["T"], max_space_length=ms["max_N"], load_meta=True, shuffle=False)
        batches.append(next(iter(ld)))
    random.seed(42)
    torch.manual_seed(1234)
    losses, objs, ns = [], [], []
    for step, batch in enumerate(batches):
        obj = random.sample(["encoding", "decoding", "token_masking"], 1)[0]
        regions = np.asarray(batch["neuron_regions"]).T
        b = dict(spikes_data=batch["spikes_data"].float(), target=batch["target"].float(), time_attn_mask=batch["time_attn_mask"],
                 spikes_timestamps=batch["spikes_timestamps"])
        out = model(make_mod_dict(b, obj, regions=regions))
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(float(out.loss.detach()))
        objs.append(obj)
        ns.append({k: int(v) for k, v in out.mod_n_examples.items()})
    print("    multisession (40 sessions) curve:", losses[:3], "...", losses[-1])
    save_json("multisession_big.json", dict(ms, neurons=neurons, loss=losses, objective=objs, n=ns,
                                            final_norm={k: float(v.double().norm()) for k, v in model.state_dict().items()}))


EVAL_DRIVER_CASES = [
    dict(mode="per_neuron", neurons=[0, 5, 9]),
    dict(mode="forward_pred", held_out_list=[5, 6, 7]),
    dict(mode="inter_region", target_regions=["CA1", "PO"], heldout_idxs=[0, 1]),
    dict(mode="intra_region", target_regions=["CA1"], heldout_idxs=[1]),
    dict(mode="modal_spike", held_out_list=[0, 1, 2, 3, 4, 5, 6, 7]),
    dict(mode="modal_behavior", held_out_list=[2, 3, 4, 5]),
]


def _eval_runner(model, batch, regions, K_, N):
    """mod_dict of the reference's evaluation loops (utils/eval_utils.py:157-193; :660-692 for modal_behavior, where the held-out mask sits on
    the behaviour channels) -> eval forward -> (targets, exp(preds) or the raw behaviour predictions, loss)."""
    def run(mask_result, mask_mode, masked="ap"):
        md = {}
        for mod in model.mod_to_indx.keys():
            md[mod] = dict(inputs_modality=torch.tensor(model.mod_to_indx[mod]), targets_modality=torch.tensor(model.mod_to_indx[mod]),
                           inputs_attn_mask=batch["time_attn_mask"], inputs_timestamp=batch["spikes_timestamps"],
                           targets_timestamp=batch["spikes_timestamps"], eid="synthetic", num_neuron=N, masking_mode=None)
            if mod == "ap":
                md[mod].update(inputs=batch["spikes_data"].clone(), inputs_regions=np.asarray([regions] * K_),
                               targets=batch["spikes_data"].clone(), mask_mode=mask_mode,
                               eval_mask=mask_result["eval_mask"] if masked == "ap" else torch.zeros_like(batch["spikes_data"]).to(torch.int64))
            else:
                md[mod].update(inputs=batch["target"].clone(), targets=batch["target"].clone(),
                               eval_mask=mask_result["eval_mask"] if masked == mod else torch.zeros_like(batch["target"]).to(torch.int64))
        with torch.no_grad():
            out = model(md)
        if masked == "ap":
            return out.mod_targets["ap"][:, :, :N].numpy(), torch.exp(out.mod_preds["ap"][:, :, :N]).numpy(), float(out.loss)
        nb = batch["target"].shape[2]
        return out.mod_targets[masked][:, :, :nb].numpy(), out.mod_preds[masked][:, :, :nb].numpy(), float(out.loss)
    return run


def _eval_cases(cases_in, run, spikes, beh, regions, heldout_mask, bps_list, arrs, keep_rates):
    """Runs every evaluation case through the reference's pieces; keep_rates(cid, tag, rates) stores what the fixture keeps of the rates."""
    cases = []
    for cid, c in enumerate(cases_in):
        out = dict(c)
        if c["mode"] == "per_neuron":
            bl = []
            for j, n_i in enumerate(c["neurons"]):
                mr = heldout_mask(spikes.clone(), mode="manual", heldout_idxs=np.array([n_i]))
                gt, pr, loss = run(mr, "neuron")
                keep_rates(cid, f"rates{j}", pr)
                bl.append(bps_list(gt[:, :, [n_i]], pr[:, :, [n_i]])[0])
            out["bps"] = bl
        elif c["mode"] in ("forward_pred", "modal_spike"):
            mr = heldout_mask(spikes.clone(), mode=c["mode"], heldout_idxs=np.array(c["held_out_list"]), target_regions=None, neuron_regions=regions)
            gt, pr, loss = run(mr, "causal")
            keep_rates(cid, "rates", pr)
            t_i = c["held_out_list"]
            out["bps"] = bps_list(gt[:, t_i], pr[:, t_i])
        elif c["mode"] == "modal_behavior":
            mr = heldout_mask(beh.clone(), mode="modal_behavior", heldout_idxs=np.array(c["held_out_list"]), target_regions=None, neuron_regions=regions)
            gt, pr, loss = run(mr, "causal", masked="behavior")
            keep_rates(cid, "rates", pr)
            out["bps"] = [float("nan")] * gt.shape[2]               # utils/eval_utils.py:709-710
        else:
            mr = heldout_mask(spikes.clone(), mode=c["mode"], heldout_idxs=np.array(c["heldout_idxs"]), target_regions=c["target_regions"],
                              neuron_regions=regions)
            gt, pr, loss = run(mr, "inter-region" if c["mode"] == "inter_region" else "intra-region")
            keep_rates(cid, "rates", pr)
            n_i = np.asarray(mr["heldout_idxs"])
            out["heldout"] = n_i.tolist()
            out["bps"] = bps_list(gt[:, :, n_i], pr[:, :, n_i])
        out["loss"] = loss
        cases.append(out)
        print("   ", c["mode"], [round(b, 4) for b in out["bps"][:4]], loss)
    return cases


def fx_eval_driver():
    """The computational core of co_smoothing_eval (utils/eval_utils.py:93-757) on synthetic trials, built from the reference's own
    pieces in the order the reference uses them: heldout_mask -> mod_dict (:157-193) -> model.eval() forward -> exp(preds) ->
    bits_per_spike on the held-out slice, neuron by neuron (:200-203, :294-303, :399-408, :511-514).  Dataset access, PSTH
    analysis and plotting (the rest of that function) are not part of the hot path and are not reproduced."""
    import datasets
    if not hasattr(datasets, "list_datasets"):
        datasets.list_datasets = lambda *a, **k: []
    _stub_wandb_torcheval()
    import matplotlib
    matplotlib.use("Agg")
    from utils.eval_utils import bits_per_spike, heldout_mask
    K_, T, N = 6, 8, 12
    model = build_model(tiny_model_cfg(), N, 2, seed=21)
    model.eval()
    g = torch.Generator().manual_seed(77)
    spikes = torch.poisson(torch.full((K_, T, N), 0.9), generator=g)
    beh = torch.randn(K_, T, 2, generator=g)
    regions = np.array(["CA1", "PO", "CA1", "LP", "PO", "CA1", "LP", "PO", "CA1", "DG", "PO", "LP"])
    batch = dict(spikes_data=spikes, target=beh, time_attn_mask=torch.ones(K_, T, dtype=torch.int64),
                 spikes_timestamps=torch.arange(T).unsqueeze(0).repeat(K_, 1))
    arrs = {f"sd/{k}": npify(v) for k, v in model.state_dict().items()}
    arrs.update(spikes=npify(spikes), behavior=npify(beh))
    arrs["regions"] = np.frombuffer(json.dumps(regions.tolist()).encode(), dtype=np.uint8)

    run = _eval_runner(model, batch, regions, K_, N)

    def bps_list(gt, pr):
        res = []
        for n_i in range(gt.shape[2]):
            with np.errstate(divide="ignore", invalid="ignore"):
                b = bits_per_spike(pr[:, :, [n_i]].astype(np.float64), gt[:, :, [n_i]].astype(np.float64))
            res.append(float("nan") if np.isinf(b) else float(b))
        return res

    def keep(cid, tag, pr):
        arrs[f"c{cid}/{tag}"] = pr.astype(np.float32)
    cases = _eval_cases(EVAL_DRIVER_CASES, run, spikes, beh, regions, heldout_mask, bps_list, arrs, keep)
    arrs["meta"] = np.frombuffer(json.dumps(dict(cases=cases, K=K_, T=T, N=N, model_seed=21)).encode(), dtype=np.uint8)
    save_npz("eval_driver.npz", **arrs)


EVAL_BIG_CASES = [
    dict(mode="per_neuron", neurons=[3, 400]),
    dict(mode="forward_pred", held_out_list=list(range(90, 100))),
    dict(mode="inter_region", target_regions=["CA1", "PO"], heldout_idxs=[0, 5, 17]),
    dict(mode="intra_region", target_regions=["LP"], heldout_idxs=[2, 3]),
    dict(mode="modal_spike", held_out_list=list(range(100))),
    dict(mode="modal_behavior", held_out_list=list(range(40, 60))),
]
EVAL_BIG = dict(K=64, T=100, N=668, model_seed=5, data_seed=177, rate_stride=(8, 10, 16))


def eval_big_inputs():
    """Synthetic test set of the big evaluation fixture, regenerated from seeds on both sides (17 MB of spikes are not committed)."""
    K_, T, N = EVAL_BIG["K"], EVAL_BIG["T"], EVAL_BIG["N"]
    g = torch.Generator().manual_seed(EVAL_BIG["data_seed"])
    base = 0.2 + 1.6 * torch.rand(N, generator=g)
    spikes = torch.poisson(base.expand(K_, T, N).contiguous(), generator=g)
    beh = torch.randn(K_, T, 2, generator=g)
    names = np.array(["CA1", "PO", "LP", "DG", "VISa"])
    regions = names[torch.randint(0, 5, (N,), generator=g).numpy()]
    return spikes, beh, regions


def fx_eval_driver_big():
    """fx_eval_driver at the size the evaluation runs at: a test set of K = 64 trials, T = 100 bins, N = 668 neurons through
    the DEFAULT model (d_model 256, 5 + 5 layers, built from a seed on both sides), every mode incl. modal_behavior.  Kept: the per-neuron
    bits/spike of every case, the loss, and a strided sample of the rates."""
    import datasets
    if not hasattr(datasets, "list_datasets"):
        datasets.list_datasets = lambda *a, **k: []
    _stub_wandb_torcheval()
    import matplotlib
    matplotlib.use("Agg")
    from utils.eval_utils import bits_per_spike, heldout_mask
    K_, T, N = EVAL_BIG["K"], EVAL_BIG["T"], EVAL_BIG["N"]
    model = build_model(ref_config().model, N, 2, seed=EVAL_BIG["model_seed"])
    model.eval()
    spikes, beh, regions = eval_big_inputs()
    batch = dict(spikes_data=spikes, target=beh, time_attn_mask=torch.ones(K_, T, dtype=torch.int64),
                 spikes_timestamps=torch.arange(T).unsqueeze(0).repeat(K_, 1))
    run = _eval_runner(model, batch, regions, K_, N)

    def bps_list(gt, pr):
        res = []
        for n_i in range(gt.shape[2]):
            with np.errstate(divide="ignore", invalid="ignore"):
                b = bits_per_spike(pr[:, :, [n_i]].astype(np.float64), gt[:, :, [n_i]].astype(np.float64))
            res.append(float("nan") if np.isinf(b) else float(b))
        return res

    arrs = {}
    sk, st_, sn = EVAL_BIG["rate_stride"]

    def keep(cid, tag, pr):
        arrs[f"c{cid}/{tag}"] = pr[::sk, ::st_, ::sn].astype(np.float32)
        arrs[f"c{cid}/{tag}_sum"] = np.asarray(pr.astype(np.float64).sum(axis=(0, 1)))
    cases = _eval_cases(EVAL_BIG_CASES, run, spikes, beh, regions, heldout_mask, bps_list, arrs, keep)
    arrs["spikes_sum"] = np.asarray(spikes.double().sum(dim=(0, 1)).numpy())          # guards the regenerated inputs
    arrs["meta"] = np.frombuffer(json.dumps(dict(cases=cases, **EVAL_BIG)).encode(), dtype=np.uint8)
    save_npz("eval_driver_big.npz", **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, fn in [("init_order", fx_init_order), ("tiny_fwd_bwd", fx_tiny_fwd_bwd),
                     ("default_scalars", fx_default_scalars), ("masker_bits", fx_masker_bits),
                     ("mask_index_ops", fx_mask_index_ops), ("sched_adamw", fx_sched_adamw),
                     ("loss_curve", fx_loss_curve), ("loss_curve_1k", fx_loss_curve_1k), ("config5_scalars", fx_config5_scalars), ("multisession_curve", fx_multisession_curve), ("eval_metrics", fx_eval_metrics), ("trainer_io", fx_trainer_io), ("loader_collate", fx_loader_collate), ("masker_modes", fx_masker_modes), ("masker_modes_model", fx_masker_modes_model), ("eval_driver", fx_eval_driver), ("eval_driver_big", fx_eval_driver_big),
                     ("h64_curve", fx_h64_curve), ("multisession_big", fx_multisession_big), ("loss_curve_1k_default", fx_loss_curve_1k_default)]:
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()


if __name__ == "__main__":
    main()

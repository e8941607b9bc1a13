"""Every launch of the B=1024 bf16 step plan with its shape, its in-order HIP-event time and its rate against the algorithmic
bytes / flops of that launch: which launches sit furthest from the ~4.2 TB/s the mixed row-strided patterns reach on this chip."""
import os, sys, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src")):
    sys.path.insert(0, p)
import torch
from torch.optim.lr_scheduler import OneCycleLR
from multi_modal_foundation_model_amd.builders import build_model, load_config
from multi_modal_foundation_model_amd.optim import make_optimizer
from multi_modal_foundation_model_amd.synthetic import synth_batch
from trainer.make import make_multimodal_trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
cfg = load_config()
model = build_model(cfg.model, 668, 2, seed=cfg.seed)
model.compute_dtype = "bf16"
model.masker.token_mask_only = True
model = model.to(dev)
opt = make_optimizer(model, lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.wd, eps=cfg.optimizer.eps)
sch = OneCycleLR(optimizer=opt, total_steps=1000, max_lr=cfg.optimizer.lr, pct_start=cfg.optimizer.warmup_pct, div_factor=cfg.optimizer.div_factor)


class Acc:
    device = dev


tr = make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=opt, log_dir="/tmp", accelerator=Acc(),
                             lr_scheduler=sch, avail_mod=["ap", "behavior"], config=cfg,
                             modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]), mixed_training=True, num_neurons=[668])
batch = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth_batch(B, 100, 668, 2, seed=0).items()}
random.seed(42); torch.manual_seed(4242)
model.train()
for i in range(3):
    tr._sample_modes()
    out = tr._forward_model_outputs(dict(batch), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
    out.loss.backward(); opt.step(); sch.step(); opt.zero_grad()
torch.cuda.synchronize()

eng = model._engine
plan = eng._last
st = torch.cuda.current_stream().cuda_stream
entries = list(plan["fwd"]) + [e for _, seg in plan["bwd"] for e in seg]
n, reps = len(entries), 3
acc = [0.0] * n
for rep in range(reps + 1):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i, (fn, args, keep) in enumerate(entries):
        evs[i].record(); fn(*args, st)
    evs[n].record(); torch.cuda.synchronize()
    if rep:
        for i in range(n): acc[i] += evs[i].elapsed_time(evs[i + 1]) / reps

groups = collections.OrderedDict()
for i, (fn, args, keep) in enumerate(entries):
    name, us = fn.__name__, acc[i] * 1e3
    by = fl = 0.0
    if name == "mmfm_gemm":
        d = keep[0]
        kind = "x.W^T" if (d.a_kcontig and d.b_kcontig) else ("dY.W" if d.a_kcontig else "dY^T.X")
        extra = ("+pre" if d.pre_out else "") + ("+res" if d.residual else "") + ("+gm" if d.gradmul_pre else "") + (f"+act{d.act}" if d.act else "") \
            + ("+drop" if d.drop.p > 0 else "") + ("+colsum" if d.colsum else "") + (f" s{d.splits}" if d.splits > 1 else "")
        key = f"gemm {kind:7s} M={d.M} N={d.N} K={d.K}{extra}"
        esz = 4 if d.c_f32 else 2
        by = 2.0 * (d.M * d.K + d.K * d.N) + esz * d.M * d.N * max(1, d.splits) + (2.0 * d.M * d.N if d.pre_out else 0) \
            + (2.0 * d.M * d.N if d.residual else 0) + (2.0 * d.M * d.N if d.gradmul_pre else 0)
        fl = 2.0 * d.M * d.N * d.K
    elif name == "mmfm_gemm_pair":
        da, db = keep[0], keep[1]
        key = f"gemm pair dY^T.X {da.M}x{da.N} s{da.splits} + {db.M}x{db.N} s{db.splits} K={da.K}"
        by = sum(2.0 * (d.M * d.K + d.K * d.N) + 4.0 * d.M * d.N * max(1, d.splits) for d in (da, db))
        fl = sum(2.0 * d.M * d.N * d.K for d in (da, db))
    elif name == "mmfm_rowgemm":
        d = keep[0]
        key = f"rowgemm R={d.R} N={d.N} K={d.K}" + (" LN" if d.ln else "") + (" +res" if d.residual else "") + (" LNbwd" if d.ln_bwd else "")
        by = 2.0 * d.R * (d.K + d.N) + (2.0 * d.R * 256 if d.residual else 0) + (2.0 * d.R * 256 if (d.ln and d.xhat) else 0) + (2.0 * d.R * 256 if d.ln_bwd else 0)
        fl = 2.0 * d.R * d.N * d.K
    elif name in ("mmfm_attn_fwd", "mmfm_attn_bwd"):
        d = keep[0]
        key = f"{name[5:]} B={d.B} h={d.heads} Lq={d.Lq} Lk={d.Lk} dh={d.dh} flags={d.flags & 0xff}"
        by = 2.0 * d.B * d.heads * d.dh * (2 * d.Lq + 2 * d.Lk) * (1 if name.endswith("fwd") else 2)
        fl = (4.0 if name.endswith("fwd") else 10.0) * d.B * d.heads * d.Lq * d.Lk * d.dh
    else:
        key = name[5:]
    g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
    g[0] += 1; g[1] += us; g[2] += by; g[3] += fl

tot = sum(g[1] for g in groups.values())
print(f"B={B}: {n} launches, {tot / 1e3:.2f} ms; fused mask {eng._fused_mask(B * 200)}")
print(f"{'launch':78s} {'n':>3s} {'us each':>8s} {'ms tot':>7s} {'TB/s':>6s} {'TF/s':>6s} {'us@4.2TB/s':>10s}")
for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    each = g[1] / g[0]
    tb = g[2] / g[1] / 1e6 if g[2] else 0
    tf = g[3] / g[1] / 1e6 if g[3] else 0
    floor = g[2] / g[0] / 4.2e6 if g[2] else 0
    print(f"{key:78s} {g[0]:3d} {each:8.1f} {g[1] / 1e3:7.2f} {tb:6.2f} {tf:6.0f} {floor:10.1f}")

import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from multi_modal_foundation_model_amd.builders import build_model, load_config
from torch.optim.lr_scheduler import OneCycleLR
from multi_modal_foundation_model_amd.optim import make_optimizer
from multi_modal_foundation_model_amd.synthetic import synth_batch
from trainer.make import make_multimodal_trainer
cfg = load_config(); dev = torch.device("cuda", 0)
model = build_model(cfg.model, 668, 2, seed=42); model.compute_dtype = "bf16"; model.masker.token_mask_only = True; model = model.to(dev)
N = 600
opt = make_optimizer(model, lr=1e-4, weight_decay=0.01, eps=1e-8)
sch = OneCycleLR(optimizer=opt, total_steps=N, max_lr=1e-4, pct_start=0.15, div_factor=10)
class Acc: device = dev
tr = make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=opt, log_dir="/tmp", accelerator=Acc(), lr_scheduler=sch,
                             avail_mod=["ap", "behavior"], config=cfg, modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]),
                             mixed_training=True, num_neurons=[668])
pool = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth_batch(256, 100, 668, 2, seed=i).items()} for i in range(8)]
random.seed(42); torch.manual_seed(0); model.train()
losses = []
for i in range(N):
    tr._sample_modes()
    out = tr._forward_model_outputs(dict(pool[i % 8]), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
    out.loss.backward(); opt.step(); sch.step(); opt.zero_grad()
    losses.append((tr.training_mode, out.loss.detach()))
vals = [(m, l.item()) for m, l in losses]
import math
assert all(math.isfinite(v) for _, v in vals)
for mode in ("encoding", "decoding", "token_masking"):
    xs = [v for m, v in vals if m == mode]
    print(mode, len(xs), "first10 mean %.4f  last10 mean %.4f" % (sum(xs[:10]) / 10, sum(xs[-10:]) / 10))

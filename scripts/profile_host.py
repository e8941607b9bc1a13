"""Host-side profile of the bench step (where does the non-GPU time go?)."""
import cProfile, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from multi_modal_foundation_model_amd.builders import build_model, load_config
from torch.optim.lr_scheduler import OneCycleLR
from multi_modal_foundation_model_amd.optim import make_optimizer
from multi_modal_foundation_model_amd.synthetic import synth_batch
from trainer.make import make_multimodal_trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = load_config()
dev = torch.device("cuda", 0)
model = build_model(cfg.model, 668, 2, seed=42)
model.compute_dtype = "bf16"
model.masker.token_mask_only = True
model = model.to(dev)
opt = make_optimizer(model, lr=1e-4, weight_decay=0.01, eps=1e-8)
sch = OneCycleLR(optimizer=opt, total_steps=1000, max_lr=1e-4, pct_start=0.15, div_factor=10)
class Acc: device = dev
tr = make_multimodal_trainer(model=model, train_dataloader=[], eval_dataloader=[], optimizer=opt, log_dir="/tmp", accelerator=Acc(), lr_scheduler=sch,
                             avail_mod=["ap", "behavior"], config=cfg, modal_filter=dict(input=["ap", "behavior"], output=["ap", "behavior"]),
                             mixed_training=True, num_neurons=[668])
pool = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth_batch(B, 100, 668, 2, seed=i).items()} for i in range(4)]
random.seed(42); model.train()
def step(i):
    tr._sample_modes()
    out = tr._forward_model_outputs(dict(pool[i % 4]), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
    out.loss.backward(); opt.step(); sch.step(); opt.zero_grad()
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(20): step(5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pr.disable()
print(f"B={B}: host issue time {1e3*(t1-t0)/20:.2f} ms/step, with final sync {1e3*(t2-t0)/20:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

"""Step time of BASELINE configs[4] as SURVEY.md 8d instantiates it (H=512, I=1024, dh=64, T=200, ap+behavior+lfp -> L=600) in
bf16 mode with dropout on, plus the per-family kernel breakdown: which kernels a config-5 run spends its time in."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from multi_modal_foundation_model_amd.builders import build_model_mods, make_optimizer, model_config
import numpy as np
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mods = [("ap", 668), ("behavior", 2), ("lfp", 128)]
model = build_model_mods(model_config(H=512, heads=8, inter=1024, max_F=200, n_modality=3), mods, seed=42)
model.loss_mod["lfp"] = "mse"
model.compute_dtype = "bf16"
model = model.cuda().train()
opt, sch = make_optimizer(model, 1000)
# synthetic batch (SURVEY.md 8d recipe: Poisson(0.3) spikes, N(0,1) for the other modalities), 'ap' fully masked
g = torch.Generator().manual_seed(0)
T = 200
attn = torch.ones(B, T, dtype=torch.int64).cuda()
ts = torch.arange(T, dtype=torch.int64)[None].repeat(B, 1).cuda()
_dev = {}
for i, (name, n) in enumerate(mods):
    x = (torch.poisson(torch.full((B, T, n), 0.3), generator=g) if name == "ap" else torch.randn(B, T, n, generator=g)).cuda()
    idx = torch.tensor(i).cuda()
    _dev[name] = dict(inputs_modality=idx, targets_modality=idx, inputs_attn_mask=attn, inputs_timestamp=ts, targets_timestamp=ts,
                      masking_mode=None, inputs=x, targets=x,
                      eval_mask=torch.full((1, 1, 1), 1 if name == "ap" else 0, dtype=torch.int64, device="cuda").expand(B, T, n))
    if name == "ap":
        _dev[name]["inputs_regions"] = np.full((B, n), "XX")
def md():                                    # inputs resident in HBM, a fresh (shallow) mod_dict per step like the trainer builds
    return {m: dict(x) for m, x in _dev.items()}
def step():
    out = model(md()); out.loss.backward(); opt.step(); sch.step(); opt.zero_grad(); return out.loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"config 5, bf16, B={B}: {dt*1e3:.1f} ms/step = {B/dt:.0f} samples/s, loss {loss.item():.4f}")
agg, subs = bench.kernel_profile(model._engine, model._engine._last, reps=1)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"   {k:24s} {v[0]:4d} launches {v[1]:8.2f} ms" + (f"  {v[2]/(v[1]*1e-3)/1e12:6.1f} TF/s" if v[2] else ""))

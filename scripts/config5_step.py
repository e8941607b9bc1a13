"""Step time of BASELINE configs[4] as SURVEY.md 8d instantiates it (H=512, I=1024, dh=64, T=200, ap+behavior+lfp -> L=600) in
bf16 mode with dropout on, plus the per-family kernel breakdown: which kernels a config-5 run spends its time in."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import build_model_mods, make_optimizer, model_config
from oracle import mm_oracle as O
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
batch = O.synth_batch_mods(B, 200, mods, seed=0)
_dev = O.make_mod_dict_mods(batch, mods, "ap")
for x in _dev.values():
    for k, v in list(x.items()):
        if isinstance(v, torch.Tensor): x[k] = v.cuda()
    x["targets_modality"], x["targets_timestamp"] = x["inputs_modality"], x["inputs_timestamp"]
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

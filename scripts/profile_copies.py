import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src")):
    sys.path.insert(0, p)
import torch
exec(open(os.path.join(ROOT, "scripts/step_launches.py")).read().split("torch.cuda.synchronize()\n\neng = model._engine")[0])
from torch.profiler import profile, ProfilerActivity
def step():
    tr._sample_modes()
    out = tr._forward_model_outputs(dict(batch), masking_mode=tr.masking_mode, training_mode=tr.training_mode)
    out.loss.backward(); opt.step(); sch.step(); opt.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
evs = [e for e in prof.events() if "copy" in e.name.lower() or "to" == e.name.split("::")[-1]]
import collections
c = collections.Counter(e.name for e in evs)
print(c.most_common(20))

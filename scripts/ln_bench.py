import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_modal_foundation_model_amd import ops
R, H = 204800, 256
x = torch.randn(R, H, device="cuda").bfloat16(); g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
y = torch.empty_like(x); mean = torch.empty(R, device="cuda"); rstd = torch.empty(R, device="cuda")
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, R, H))
print(os.environ.get("MMFM_LN_FWD_UNR"), os.environ.get("MMFM_LN_FWD_BLOCKS"), f"{us:.1f} us {2*R*H*2/us/1e6:.2f} TB/s")

from multi_modal_foundation_model_amd import _lib as Lb
dy = torch.randn(R, H, device="cuda").bfloat16(); dres = torch.randn(R, H, device="cuda").bfloat16(); dx = torch.empty_like(x)
dg, db = torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
ws = torch.empty(Lb.lib().mmfm_layernorm_bwd_workspace(R, H) // 4, device="cuda")
us = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres, dx, dg, db, R, H, ws))
print("bwd", os.environ.get("MMFM_LN_BWD_UNR"), os.environ.get("MMFM_LN_BWD_BLOCKS"), f"{us:.1f} us {4*R*H*2/us/1e6:.2f} TB/s")

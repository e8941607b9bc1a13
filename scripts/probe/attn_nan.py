import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops, _lib as Lb
B, heads, Lq, Lk, dh, p = 2, 8, 200, 200, 32, 0.4
H = heads * dh
for pad, flags, scale6 in ((False, 0, False), (True, 0, False), (True, 1, False), (False, 0, True), (True, 1, True)):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * Lq, H, generator=g).cuda().to(torch.bfloat16)
    kv = torch.randn(B * Lk, 2 * H, generator=g).cuda()
    if scale6: kv.view(B, Lk, 2 * H)[:, 150:, :H] *= 6.0
    kv = kv.to(torch.bfloat16)
    kp = torch.ones(B, Lk, dtype=torch.uint8)
    if pad: kp[0, Lk - 3:] = 0; kp[B - 1, 5:9] = 0
    kp = kp.cuda()
    state = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(state, 4321)
    dmask = torch.zeros(ops.attn_dropmask_bytes(B, heads, Lq, Lk) // 4, dtype=torch.int32, device="cuda")
    o, lse = torch.empty(B * Lq, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, Lq, device="cuda")
    desc = ops.attn_desc(Lb.BF16, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, flags, dh ** -0.5, drop_p=ops.dropout(state, 7, p), drop_mask=dmask)
    ops.attn_fwd(desc); torch.cuda.synchronize()
    bad = torch.isnan(o.float()).view(B, Lq, heads, dh).any(-1)
    print(f"pad={pad} flags={flags} scale6={scale6}: nan rows {int(bad.sum())} lse nan {int(torch.isnan(lse).sum())}", bad.nonzero()[:12].tolist())

"""debug: fast attention forward/backward vs torch on small shapes"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import _lib as Lb, ops
torch.manual_seed(0)
def run(B, heads, L, p=0.0, pad=False, flags=0):
    dh = 32; H = heads * dh
    qkv = torch.randn(B * L, 3 * H).cuda().bfloat16()
    d_o = torch.randn(B * L, H).cuda().bfloat16()
    kp = torch.ones(B, L, dtype=torch.uint8)
    if pad: kp[0, L - 3:] = 0
    kp = kp.cuda()
    o, lse = torch.zeros(B * L, H, device="cuda", dtype=torch.bfloat16), torch.zeros(B, heads, L, device="cuda")
    dqkv = torch.zeros(B * L, 3 * H, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(st, 5)
    kb = torch.zeros(ops.attn_keepbits_bytes(B, heads, L, L), dtype=torch.uint8, device="cuda") if p > 0 else None
    base = qkv.data_ptr()
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * 2, base + 4 * H, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None, flags,
                         1 / math.sqrt(dh), drop_p=ops.dropout(st, 3, p), d_o=d_o.data_ptr(), lddo=H, dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + 2 * H,
                         dv=dqkv.data_ptr() + 4 * H, lddq=3 * H, lddk=3 * H, lddv=3 * H, keepbits=kb)
    ops.attn_fwd(desc); torch.cuda.synchronize()
    x = qkv.float().requires_grad_(True)
    q, k, v = [t.view(B, L, heads, dh).transpose(1, 2) for t in x.split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    m = kp.bool()[:, None, None, :].expand(B, heads, L, L)
    if flags & 1: m = m | torch.eye(L, dtype=torch.bool, device="cuda")[None, None]
    s = s.masked_fill(~m, float("-inf"))
    lref = torch.logsumexp(s, -1)
    print(f"B{B} h{heads} L{L} p{p} pad{pad} fl{flags}: lse err {(lse - lref).abs().max().item():.3e}", end="  ")
    if p == 0:
        oref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H)
        e = (o.float() - oref).abs()
        print(f"o err {e.max().item():.3e} at row {e.max(1).values.argmax().item()} col {e.max(0).values.argmax().item()}", end="  ")
        ops.attn_bwd(desc); torch.cuda.synchronize()
        oref.backward(d_o.float())
        for nm, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
            print(f"{nm} {(dqkv[:, sl].float() - x.grad[:, sl]).abs().max().item():.3e}/{x.grad[:, sl].abs().max().item():.2e}", end=" ")
    print()
for L in (32, 64, 72, 200, 224):
    run(1, 1, L)
run(2, 8, 200)
run(2, 8, 200, pad=True)
run(2, 8, 200, pad=True, flags=1)
run(2, 8, 200, p=0.4)

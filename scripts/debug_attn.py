import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from multi_modal_foundation_model_amd import _lib as Lb, ops

def run(B, heads, L, dh, flags, pad):
    H = heads * dh
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * L, 3 * H, generator=g).cuda().to(torch.bfloat16)
    keypad = torch.ones(B, L, dtype=torch.uint8)
    if pad:
        keypad[0, L - 3:] = 0
    kp = keypad.cuda()
    o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
    base = qkv.data_ptr()
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * 2, base + 2 * H * 2, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None,
                         flags, 1 / math.sqrt(dh))
    ops.attn_fwd(desc)
    m = kp.bool()[:, None, :].expand(B, L, L)
    if flags & 1:
        m = m | torch.eye(L, dtype=torch.bool, device="cuda")[None]
    x = qkv.float()
    q, k, v = [t.view(B, L, heads, dh).transpose(1, 2) for t in x.split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(~m[:, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H)
    err = (o.float() - ref).abs().view(B, L, heads, dh)
    lref = torch.logsumexp(s, -1)
    print(f"B{B} h{heads} L{L} dh{dh} flags{flags} pad{pad}: max err {err.max().item():.4f}; per-b {err.amax((1,2,3)).tolist()}; "
          f"per-qtile {[round(err[:, i:i+32].max().item(),3) for i in range(0, L, 32)]}; lse err {(lse-lref).abs().max().item():.4f}")

for args in [(2, 8, 200, 32, 0, False), (2, 8, 200, 32, 0, True), (2, 8, 64, 32, 0, False), (2, 8, 64, 32, 0, True), (1, 2, 32, 32, 0, False),
             (2, 2, 70, 64, 1, True), (2, 2, 64, 64, 0, False), (2, 4, 48, 16, 0, True)]:
    run(*args)

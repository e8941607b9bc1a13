"""Phase breakdown of the fast attention forward from the stamped diagnostic library (scripts/probe/build_attn_stamp.sh)."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MMFM_LIB"] = os.path.join(ROOT, "multi_modal_foundation_model_amd", "libmmfm_astamp.so")
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops, _lib as Lb
B, heads, L, dh = 1024, 8, 200, 32
H = heads * dh
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * L, 3 * H, generator=g).cuda().to(torch.bfloat16)
kp = torch.ones(B, L, dtype=torch.uint8, device="cuda")
o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
state = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(state, 7)
lib = Lb.lib()
lib.mmfm_attn_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
names = ["prologue: K/V/bias -> LDS", "barrier + vote", "Q operands, first score tile", "key-tile loop", "epilogue (normalise, stage, store)"]
for p in (0.0, 0.4):
    base = qkv.data_ptr()
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * 2, base + 2 * H * 2, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None, 1,
                         1 / math.sqrt(dh), drop_p=ops.dropout(state, 3, p) if p else None, drop_o=ops.dropout(state, 4, p) if p else None)
    ops.attn_fwd(desc); torch.cuda.synchronize(); lib.mmfm_attn_probe_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attn_fwd(desc); e1.record(); torch.cuda.synchronize()
    lib.mmfm_attn_probe_read(buf, 1)
    nw = B * heads * 7
    tot = sum(buf[i] for i in range(5))
    print(f"attn_fwd_fast p={p}: {e0.elapsed_time(e1)*1e3:.1f} us (stamped build); cycles per wave {tot/nw:.0f}")
    for i, n in enumerate(names):
        print(f"   {n:40s} {buf[i]/nw:10.0f}  ({100*buf[i]/tot:4.1f} %)")

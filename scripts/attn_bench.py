"""Standalone timing of the attention kernels at the step's shapes (for rocprofv3 / PMC runs)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import _lib as Lb, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
keepbits = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # 1: keep-bit dropout path (round 4), 0: counter hash in both directions
heads, L, dh = 8, 200, 32
H = heads * dh
td = torch.bfloat16 if dtype == "bf16" else torch.float32
es = 2 if dtype == "bf16" else 4
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * L, 3 * H, generator=g).cuda().to(td)
d_o = torch.randn(B * L, H, generator=g).cuda().to(td)
kp = torch.ones(B, L, dtype=torch.uint8, device="cuda")
o, lse = torch.empty(B * L, H, device="cuda", dtype=td), torch.empty(B, heads, L, device="cuda")
dqkv = torch.empty(B * L, 3 * H, device="cuda", dtype=td)
state = torch.zeros(2, dtype=torch.int32, device="cuda")
ops.rng_seed(state, 7)
base = qkv.data_ptr()
kb = torch.empty(ops.attn_keepbits_bytes(B, heads, L, L), dtype=torch.uint8, device="cuda") if (keepbits and dtype == "bf16" and p > 0) else None
desc = ops.attn_desc(Lb.BF16 if dtype == "bf16" else Lb.F32, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp,
                     None, 1, 1 / math.sqrt(dh), drop_p=ops.dropout(state, 3, p), drop_o=ops.dropout(state, 4, p), d_o=d_o.data_ptr(), lddo=H,
                     dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + H * es, dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H, keepbits=kb)
for fn, name, fl in ((ops.attn_fwd, "fwd", 4.0), (ops.attn_bwd, "bwd", 10.0)):
    fn(desc); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(desc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"attn_{name} {dtype} B={B} p={p} keepbits={int(kb is not None)}: {ms*1e3:.1f} us  {fl*B*heads*L*L*dh/ms/1e9:.1f} TFLOP/s")

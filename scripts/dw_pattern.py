"""Access-pattern experiment for the streaming dW kernel: one 128 x 128 tile, 256 K-slabs (every CU streams its own rows), with the operand rows
(a) dense 256-B rows, (b) 256-B segments of 1536-B rows.  Reports operand TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops
reps = 10
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
import sys
R = 204800 * int(sys.argv[1]) if len(sys.argv) > 1 else 204800 * 2
for ld in (128, 256, 768):
    for S in (256,):
        dy = (torch.randn(R, ld, device="cuda") * 0.5).to(torch.bfloat16)
        x = (torch.randn(R, ld, device="cuda") * 0.5).to(torch.bfloat16)
        kchunk = (-(-R // S) + 63) // 64 * 64
        S2 = -(-R // kchunk)
        stride = 128 * 128 + 128
        slabs = torch.empty(S2, stride, device="cuda")
        g = t(lambda: ops.gemm(dy, x, slabs, 128, 128, R, lda=ld, ldb=ld, ldc=128, a_kcontig=0, b_kcontig=0, splits=S2, kchunk=kchunk, slab_stride=stride, c_f32=1,
                               colsum=slabs.data_ptr() + 4 * 128 * 128))
        print(f"ld {ld:4d} S {S2:4d}: {g*1e3:7.1f} us  {R*256*2/g/1e9:6.2f} TB/s")

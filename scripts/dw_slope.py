"""Streaming rate vs fixed cost of the dW kernel: time at R, 2R, 4R rows with the split count held (slope = per-byte rate, intercept = launch +
ring fill + slab store)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops, _lib as L
reps = 10
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, N, K in [("qkv 768x256", 768, 256), ("proj 256x256", 256, 256), ("down 256x512", 256, 512)]:
    tiles = L.lib().mmfm_gemm_dw_tiles(N, K, 204800)
    S0 = 256 // tiles
    pts = []
    for R in (51200, 102400, 204800, 409600):
        kchunk = (-(-R // S0) + 63) // 64 * 64
        S = -(-R // kchunk)
        dy, x = (torch.randn(R, N, device="cuda") * 0.5).to(torch.bfloat16), (torch.randn(R, K, device="cuda") * 0.5).to(torch.bfloat16)
        stride = (N * K + N + 7) // 8 * 8
        slabs = torch.empty(S, stride, device="cuda")
        g = t(lambda: ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk, slab_stride=stride, c_f32=1,
                               colsum=slabs.data_ptr() + 4 * N * K))
        pts.append((R, g * 1e3))
        del dy, x
    (r0, t0), (r1, t1) = pts[1], pts[3]
    slope = (t1 - t0) / (r1 - r0)
    print(f"{name}: S {S0} " + "  ".join(f"R={r}: {u:.1f} us" for r, u in pts) + f"   slope -> {(N + K) * 2 / slope / 1e6:.2f} TB/s, intercept {t0 - slope * r0:.1f} us")

"""dW = dY^T X (+ column sums) timing at the bench-batch shapes: the streaming kernel (csrc/gemm_dw.hip, one (tile, K-slab) item per CU)
against the general 128-tile kernel (MMFM_GEMM_DW=0 in a second process) with the split count the engine picks for each, slab
reduction included.  Random operands (cdna_hip_programming.md rule 25).  usage: dw_bench.py [splits-override]"""
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
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
new = os.environ.get("MMFM_GEMM_DW", "1") != "0"
tag = "stream" if new else "128-tile"
over = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print(f"{tag}: {'shape':24s} {'S':>4s} {'gemm us':>8s} {'reduce us':>9s} {'TB/s (operands once)':>8s}")
for name, R, N, K in [("qkv 768x256", 204800, 768, 256), ("up 512x256", 204800, 512, 256), ("down 256x512", 204800, 256, 512), ("proj 256x256", 204800, 256, 256),
                      ("tok-out 256x1336", 102400, 256, 1336)]:
    tiles = -(-N // 128) * -(-K // 128)
    if new:
        from multi_modal_foundation_model_amd import _lib as L
        tiles = L.lib().mmfm_gemm_dw_tiles(N, K, 204800)
    S = over or (max(1, min(256 // tiles, R // 512)) if new else max(1, min(R // 512, max(1, 768 // tiles), 128)))
    kchunk = (-(-R // S) + 63) // 64 * 64
    S = -(-R // kchunk)
    dy, x = rnd(R, N), rnd(R, K)
    stride = (N * K + N + 7) // 8 * 8
    slabs = torch.empty(S, stride, device="cuda")
    out = torch.empty(N * K + N, device="cuda")
    g = t(lambda: ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk, slab_stride=stride, c_f32=1,
                           colsum=slabs.data_ptr() + 4 * N * K))
    r = t(lambda: ops.reduce_slabs(out, slabs, N * K + N, S, stride))
    ref = dy[:4096].double().T @ x[:4096].double()
    print(f"{tag}: {name:24s} {S:4d} {g*1e3:8.1f} {r*1e3:9.1f} {R*(N+K)*2/g/1e9:8.2f}")

"""Per-shape timing of mmfm_gemm at the step's shapes (bf16 or fp32): TFLOP/s and GB/s next to the roofline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
td = torch.bfloat16 if dtype == "bf16" else torch.float32
es = 2 if dtype == "bf16" else 4
R, BT = B * 200, B * 100
reps = 10

def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def rnd(*s): return (torch.randn(*s, device="cuda") * 0.1).to(td)

print(f"{'kind':6s} {'M':>7s} {'N':>5s} {'K':>7s} {'us':>8s} {'TF/s':>7s} {'GB/s':>7s}")
for name, M, N, K in [("qkv", R, 768, 256), ("proj", R, 256, 256), ("up", R, 512, 256), ("down", R, 256, 512), ("tok", BT, 1336, 668),
                      ("eproj", BT, 256, 1336), ("head", BT, 668, 256)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    y, res = torch.empty(M, N, device="cuda", dtype=td), rnd(M, N)
    ms = t(lambda: ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b))
    by = (M * K + N * K + M * N) * es
    print(f"NT {name:5s} {M:7d} {N:5d} {K:7d} {ms*1e3:8.1f} {2*M*N*K/ms/1e9:7.1f} {by/ms/1e6:7.0f}")
    # dX = dY[M,N] @ W[N,K]
    dy, dx = rnd(M, N), torch.empty(M, K, device="cuda", dtype=td)
    ms = t(lambda: ops.gemm(dy, w, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kcontig=0))
    print(f"NN {name:5s} {M:7d} {K:5d} {N:7d} {ms*1e3:8.1f} {2*M*N*K/ms/1e9:7.1f} {by/ms/1e6:7.0f}")
    # dW[N,K] = dY^T X, split-K like the engine
    tiles = -(-N // 128) * -(-K // 128)
    SLOTS = int(os.environ.get("DW_SLOTS", "512"))
    S = max(1, min(M // 512, -(-SLOTS // tiles) if SLOTS == 512 else max(1, SLOTS // tiles)))
    kchunk = (-(-M // S) + 63) // 64 * 64
    S = -(-M // kchunk)
    slabs, dw = torch.empty(S, N, K, device="cuda"), torch.empty(N, K, device="cuda")
    ms = t(lambda: ops.gemm(dy, x, slabs, N, K, M, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk, slab_stride=N * K, c_f32=1))
    ms2 = t(lambda: ops.reduce_slabs(dw, slabs, N * K, S, N * K))
    print(f"TN {name:5s} {N:7d} {K:5d} {M:7d} {ms*1e3:8.1f} {2*M*N*K/ms/1e9:7.1f} {(M*K+M*N)*es/ms/1e6:7.0f}   S={S} reduce {ms2*1e3:.1f} us")

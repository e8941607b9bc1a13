"""Reference point only (not used by the product): what does the vendor library (hipBLASLt via torch.mm / addmm) reach on the
step's GEMM shapes?  Tells how far the hand-written kernel is from a tuned one on the same machine."""
import sys, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R, BT = B * 200, B * 100
td = torch.bfloat16
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, M, N, K in [("qkv", R, 768, 256), ("proj", R, 256, 256), ("up", R, 512, 256), ("down", R, 256, 512), ("tok", BT, 1336, 668), ("head", BT, 668, 256)]:
    x, w, b = torch.randn(M, K, device="cuda", dtype=td), torch.randn(N, K, device="cuda", dtype=td), torch.randn(N, device="cuda", dtype=td)
    dy = torch.randn(M, N, device="cuda", dtype=td)
    us = t(lambda: torch.addmm(b, x, w.t()))
    us2 = t(lambda: torch.mm(dy, w))
    us3 = t(lambda: torch.mm(dy.t(), x))
    f = 2 * M * N * K
    print(f"{name:5s} NT(addmm) {us:7.1f} us {f/us/1e6:6.0f} TF | NN {us2:7.1f} us {f/us2/1e6:6.0f} TF | TN {us3:7.1f} us {f/us3/1e6:6.0f} TF")

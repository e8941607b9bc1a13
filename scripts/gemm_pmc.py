import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops
td = torch.bfloat16
R = 1024 * 200
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.1).to(td)
for name, M, N, K in [("proj", R, 256, 256), ("qkv", R, 768, 256), ("down", R, 256, 512)]:
    x, w, b = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=td)
    for _ in range(3):
        ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b)
    torch.cuda.synchronize()

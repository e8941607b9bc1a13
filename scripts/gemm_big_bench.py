"""x.W^T timing at the compute-bound shapes (token embedding, d_model-512 linears): the 256-tile kernel (csrc/gemm_big.hip) against the
128-tile kernel (MMFM_GEMM_BIG=0 in a second process).  Random operands (cdna_hip_programming.md rule 25)."""
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
tag = "256-tile" if os.environ.get("MMFM_GEMM_BIG", "1") != "0" else "128-tile"
print(f"{tag}: {'shape':28s} {'us':>8s} {'TF/s':>7s}")
for name, M, N, K, kw in [("tok 668->1336 (K pad 704)", 102400, 1336, 704, dict(act=2, pre=True)), ("tok 1336->256 (K pad 1344)", 102400, 256, 1344, dict(drop=True)),
                          ("cfg5 qkv", 153600, 1536, 512, {}), ("cfg5 up", 153600, 1024, 512, dict(act=1)), ("cfg5 down", 153600, 512, 1024, dict(res=True)),
                          ("cfg5 proj", 153600, 512, 512, dict(res=True)), ("square 4096", 4096, 4096, 4096, {}), ("square 8192", 8192, 8192, 8192, {})]:
    x, w, b = rnd(M, K), rnd(N, K) * (K ** -0.5), torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty_like(y) if kw.get("pre") else None
    res = rnd(M, N) if kw.get("res") else None
    st = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(st, 1)
    drop = ops.dropout(st, 3, 0.2) if kw.get("drop") else None
    ms = t(lambda: ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=kw.get("act", 0), pre_out=pre, residual=res, ldr=N, drop=drop))
    print(f"{tag}: {name:28s} {ms*1e3:8.1f} {2*M*N*K/ms/1e9:7.1f}")

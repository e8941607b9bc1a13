"""Do two independent GEMMs of a layer's backward (dW = dY^T X split-K and dX = dY W) overlap when issued on two streams?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
td = torch.bfloat16
R = B * 200
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.1).to(td)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def bench(fa, fb, reps=20):
    def run(par):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            if par:
                ev = torch.cuda.Event(); ev.record()
                s1.wait_event(ev); s2.wait_event(ev)
                with torch.cuda.stream(s1): fa()
                with torch.cuda.stream(s2): fb()
                torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
            else:
                fa(); fb()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    run(False); run(True)
    return run(False), run(True)

for name, N, K in [("qkv", 768, 256), ("proj", 256, 256), ("up", 512, 256), ("down", 256, 512)]:
    M = R
    x, w = rnd(M, K), rnd(N, K)
    dy, dx = rnd(M, N), torch.empty(M, K, device="cuda", dtype=td)
    tiles = -(-N // 128) * -(-K // 128)
    S = max(1, min(M // 512, -(-512 // tiles)))
    kchunk = (-(-M // S) + 63) // 64 * 64
    S = -(-M // kchunk)
    slabs = torch.empty(S, N, K, device="cuda")
    f_dx = lambda: ops.gemm(dy, w, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kcontig=0)
    f_dw = lambda: ops.gemm(dy, x, slabs, N, K, M, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk, slab_stride=N * K, c_f32=1)
    seq, par = bench(f_dx, f_dw)
    print(f"{name:5s} dX+dW sequential {seq:7.1f} us   two streams {par:7.1f} us   ratio {par/seq:.2f}")
# forward GEMM next to an attention-like VALU-bound kernel is not independent in the model; only the backward pair is.

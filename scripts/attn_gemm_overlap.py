"""Do a VALU-bound attention kernel and a memory/latency-bound dW GEMM overlap when issued on two streams?"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import _lib as Lb, ops

B, heads, L, dh = 1024, 8, 200, 32
H, td, es = heads * dh, torch.bfloat16, 2
R = B * L
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * L, 3 * H, generator=g).cuda().to(td)
d_o = torch.randn(B * L, H, generator=g).cuda().to(td)
kp = torch.ones(B, L, dtype=torch.uint8, device="cuda")
o, lse = torch.empty(B * L, H, device="cuda", dtype=td), torch.empty(B, heads, L, device="cuda")
dqkv = torch.empty(B * L, 3 * H, device="cuda", dtype=td)
state = torch.zeros(2, dtype=torch.int32, device="cuda")
ops.rng_seed(state, 7)
base = qkv.data_ptr()
desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None, 1,
                     1 / math.sqrt(dh), drop_p=ops.dropout(state, 3, 0.4), drop_o=ops.dropout(state, 4, 0.4), d_o=d_o.data_ptr(), lddo=H,
                     dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + H * es, dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
ops.attn_fwd(desc)
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.1).to(td)
N, K = 512, 256
x, dy = rnd(R, K), rnd(R, N)
tiles = Lb.lib().mmfm_gemm_dw_tiles(N, K, R)
S = max(1, min(R // 512, 256 // tiles))
kchunk = (-(-R // S) + 63) // 64 * 64
S = -(-R // kchunk)
slabs = torch.empty(S, N * K + N, device="cuda")
w, dx = rnd(N, K), torch.empty(R, K, device="cuda", dtype=td)
f_attn = lambda: ops.attn_bwd(desc)
f_dw = lambda: [ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk, slab_stride=N * K + N, c_f32=1, colsum=slabs.data_ptr() + 4 * N * K) for _ in range(4)]
f_dx = lambda: [ops.gemm(dy, w, dx, R, K, N, lda=N, ldb=K, ldc=K, b_kcontig=0) for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def bench(fa, fb, reps=10):
    def run(par):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            if par:
                ev = torch.cuda.Event(); ev.record()
                s1.wait_event(ev); s2.wait_event(ev)
                if os.environ.get("OVL_DW_FIRST"):
                    with torch.cuda.stream(s2): fb()
                    with torch.cuda.stream(s1): fa()
                else:
                    with torch.cuda.stream(s1): fa()
                    with torch.cuda.stream(s2): fb()
                torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
            else:
                fa(); fb()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    run(False); run(True)
    return run(False), run(True)
for name, fb in (("4 x dW(up)", f_dw), ("4 x dX(up)", f_dx)):
    seq, par = bench(f_attn, fb)
    print(f"attn_bwd + {name}: sequential {seq:7.1f} us   two streams {par:7.1f} us   ratio {par/seq:.2f}")

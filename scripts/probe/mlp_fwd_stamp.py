"""Phase breakdown of the wave-pair MLP forward (DMA ring) from the stamped diagnostic library (scripts/probe/build_stamp.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MMFM_LIB"] = os.path.join(ROOT, "multi_modal_foundation_model_amd", "libmmfm_stamp.so")
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops, _lib as L
R = 204800
BF = torch.bfloat16
def rnd(*s, sc=1.0): return torch.randn(*s, device="cuda") * sc
def prep(W, g=None, b=None, bias=None):
    N, K = W.shape
    e = dict(W=W, gamma=g, beta=b, bias=bias, Wp=torch.empty(N, K, device="cuda", dtype=BF), WpT=torch.empty(K, N, device="cuda", dtype=BF), bp=torch.empty(N, device="cuda"),
             WpP=torch.empty(N, K, device="cuda", dtype=BF), WpTP=torch.empty(K, N, device="cuda", dtype=BF))
    tb, n, tiles = ops.prep_table([e], "cuda"); ops.prep_weights(tb, n, tiles); return e
g, bt = 1 + 0.1 * rnd(256), 0.1 * rnd(256)
x = rnd(R, 256).to(BF)
xhat, rstd = torch.empty(R, 256, device="cuda", dtype=BF), torch.empty(R, device="cuda")
up, dn = prep(rnd(512, 256, sc=1 / 16), g, bt, 0.1 * rnd(512)), prep(rnd(256, 512, sc=1 / 22), None, None, 0.1 * rnd(256))
y = torch.empty(R, 256, device="cuda", dtype=BF)
lib = L.lib()
lib.mmfm_probe_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
names = ["pass prologue (x rows, LayerNorm, x_hat out)", "ring step 1 (wait, barrier, DMA issue)", "up MFMAs", "bias + GELU + cvt + exchange write",
         "ring step 2 (wait, barrier, DMA issue)", "exchange read + down MFMAs", "pass epilogue (bias, dropout, residual, y out)", "-"]
for p in (0.0, 0.4):
    st = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(st, 1)
    d = ops.mlp_desc(R, x=x, w_up=up["Wp"], b_up=up["bp"], w_down=dn["WpP"], b_down=dn["bp"], y=y, xhat=xhat, rstd=rstd,
                     drop=ops.dropout(st, 3, p) if p else None)
    ops.mlp_fwd(d); torch.cuda.synchronize(); lib.mmfm_probe_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.mlp_fwd(d); e1.record(); torch.cuda.synchronize()
    lib.mmfm_probe_read(buf, 1)
    tot = sum(buf[i] for i in range(8))
    print(f"mlp_fwd p={p}: {e0.elapsed_time(e1)*1e3:.1f} us; wave-0 cycles per workgroup (256 WGs), total {tot/256:.0f}")
    for i, n in enumerate(names[:7]):
        print(f"   {n:52s} {buf[i]/256:10.0f}  ({100*buf[i]/tot:4.1f} %)")

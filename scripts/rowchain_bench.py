"""Row-owner fused kernels at the step's shapes next to the un-fused launches they replace (bf16, R = B*200)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ROT = int(os.environ.get("ROT", "1"))
R = B * 200
BF = torch.bfloat16
reps = 10


def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def rnd(*s, sc=1.0): return (torch.randn(*s, device="cuda") * sc)


def prep(W, g=None, b=None, bias=None):
    N, K = W.shape
    e = dict(W=W, gamma=g, beta=b, bias=bias, Wp=torch.empty(N, K, device="cuda", dtype=BF), WpT=torch.empty(K, N, device="cuda", dtype=BF),
             bp=torch.empty(N, device="cuda"), WpP=torch.empty(N, K, device="cuda", dtype=BF), WpTP=torch.empty(K, N, device="cuda", dtype=BF))
    tb, n, tiles = ops.prep_table([e], "cuda")
    ops.prep_weights(tb, n, tiles)
    return e


x = rnd(R, 256).to(BF)
g, bt = 1 + 0.1 * rnd(256), 0.1 * rnd(256)
mean, rstd = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
h = torch.empty(R, 256, device="cuda", dtype=BF)
xhat = torch.empty(R, 256, device="cuda", dtype=BF)
print(f"B={B} R={R}")
for name, N in (("qkv", 768), ("kv/up-like", 512), ("q/ctx", 256)):
    W, bias = rnd(N, 256, sc=1 / 16), rnd(N)
    e = prep(W, g, bt, bias)
    Wb = W.to(BF)
    y = torch.empty(R, N, device="cuda", dtype=BF)
    t_ln = t(lambda: ops.layernorm_fwd(x, g, bt, h, mean, rstd, R, 256))
    t_g = t(lambda: ops.gemm(h, Wb, y, R, N, 256, lda=256, ldb=256, ldc=N, bias=bias))
    t_f = t(lambda: ops.rowgemm(x, e["Wp"], y, R, N, 256, bias=e["bp"], ln=True, xhat=xhat, rstd=rstd, stream_out=True, rotate=ROT))
    by = (R * 256 * 2 + R * N) * 2
    print(f"LN+{name:10s} N={N}: un-fused {t_ln:6.1f} + {t_g:6.1f} = {t_ln + t_g:6.1f} us   fused {t_f:6.1f} us  ({by / t_f / 1e6:.2f} TB/s, {2 * R * N * 256 / t_f / 1e6:.0f} TF/s)")

W, bias, res = rnd(256, 256, sc=1 / 16), rnd(256), rnd(R, 256).to(BF)
Wb = W.to(BF)
y = torch.empty(R, 256, device="cuda", dtype=BF)
t_g = t(lambda: ops.gemm(x, Wb, y, R, 256, 256, lda=256, ldb=256, ldc=256, bias=bias, residual=res, ldr=256))
t_f = t(lambda: ops.rowgemm(x, Wb, y, R, 256, 256, bias=bias, residual=res, ldr=256, rotate=ROT))
print(f"out_proj+res       : gemm {t_g:6.1f} us   rowgemm {t_f:6.1f} us  ({R * 256 * 2 * 3 / t_f / 1e6:.2f} TB/s)")

for K in (768, 512, 256):
    dy = rnd(R, K).to(BF)
    W = rnd(K, 256, sc=K ** -0.5)
    e = prep(W, g)
    dx, dres = torch.empty(R, 256, device="cuda", dtype=BF), rnd(R, 256).to(BF)
    dh = torch.empty(R, 256, device="cuda", dtype=BF)
    ws = torch.empty(max(1, ops.L.lib().mmfm_layernorm_bwd_workspace(R, 256) // 4), device="cuda")
    dg, db = torch.empty(256, device="cuda"), torch.empty(256, device="cuda")
    t_g = t(lambda: ops.gemm(dy, e["Wp"], dh, R, 256, K, lda=K, ldb=256, ldc=256, b_kcontig=0))
    t_l = t(lambda: ops.layernorm_bwd(dh, x, mean, rstd, g, dres, dx, dg, db, R, 256, ws))
    t_f = t(lambda: ops.rowgemm(dy, e["WpT"], dx, R, 256, K, ldw=K, residual=dres, ldr=256, ln_bwd=True, bwd_xhat=xhat, bwd_rstd=rstd))
    by = (R * K + 3 * R * 256) * 2
    print(f"dX+LNbwd K={K:4d}    : un-fused {t_g:6.1f} + {t_l:6.1f} = {t_g + t_l:6.1f} us   fused {t_f:6.1f} us  ({by / t_f / 1e6:.2f} TB/s, {2 * R * K * 256 / t_f / 1e6:.0f} TF/s)")

Wu, bu, Wd, bd = rnd(512, 256, sc=1 / 16), 0.1 * rnd(512), rnd(256, 512, sc=1 / 22), 0.1 * rnd(256)
up, dn = prep(Wu, g, bt, bu), prep(Wd, None, None, bd)
state = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(state, 1)
drop = ops.dropout(state, 3, 0.4)
u, gg, y = torch.empty(R, 512, device="cuda", dtype=BF), torch.empty(R, 512, device="cuda", dtype=BF), torch.empty(R, 256, device="cuda", dtype=BF)
Wub, Wdb = Wu.to(BF), Wd.to(BF)
t_ln = t(lambda: ops.layernorm_fwd(x, g, bt, h, mean, rstd, R, 256))
t_up = t(lambda: ops.gemm(h, Wub, gg, R, 512, 256, lda=256, ldb=256, ldc=512, bias=bu, pre_out=u, act=1))
t_dn = t(lambda: ops.gemm(gg, Wdb, y, R, 256, 512, lda=512, ldb=512, ldc=256, bias=bd, drop=drop, residual=x, ldr=256))
for p, dr in (("p=0.4", drop), ("p=0", None)):
    d = ops.mlp_desc(R, x=x, w_up=up["Wp"], b_up=up["bp"], w_down=dn["WpP"], b_down=dn["bp"], y=y, xhat=xhat, rstd=rstd, drop=dr, rotate=ROT)
    t_f = t(lambda: ops.mlp_fwd(d))
    print(f"MLP fwd {p:6s}     : un-fused {t_ln:6.1f} + {t_up:6.1f} + {t_dn:6.1f} = {t_ln + t_up + t_dn:6.1f} us   fused {t_f:6.1f} us  ({4 * R * 256 * 512 / t_f / 1e6:.0f} TF/s)")
dy = rnd(R, 256).to(BF)
t1, du, dx = torch.empty(R, 256, device="cuda", dtype=BF), torch.empty(R, 512, device="cuda", dtype=BF), torch.empty(R, 256, device="cuda", dtype=BF)
for p, dr in (("p=0.4", drop), ("p=0", None)):
    d = ops.mlp_desc(R, w_up=up["Wp"], b_up=up["bp"], drop=dr, xhat=xhat, rstd=rstd, dy=dy, w_down_t=dn["WpT"], w_up_t=up["WpTP"], t1=t1, g=gg, du=du, dx=dx, rotate=ROT)
    t_f = t(lambda: ops.mlp_bwd(d))
    print(f"MLP bwd {p:6s} (dX chain: recompute + dg + dh + LN bwd): fused {t_f:6.1f} us  ({6 * R * 256 * 512 / t_f / 1e6:.0f} TF/s)")

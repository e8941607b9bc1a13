"""A few launches of each row-owner kernel at the step's shapes, for rocprofv3 --pmc runs (scripts/rowchain_sq.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_modal_foundation_model_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = B * 200
BF = torch.bfloat16
def rnd(*s, sc=1.0): return torch.randn(*s, device="cuda") * sc
def prep(W, g=None, b=None, bias=None):
    N, K = W.shape
    e = dict(W=W, gamma=g, beta=b, bias=bias, Wp=torch.empty(N, K, device="cuda", dtype=BF), WpT=torch.empty(K, N, device="cuda", dtype=BF), bp=torch.empty(N, device="cuda"),
             WpP=torch.empty(N, K, device="cuda", dtype=BF), WpTP=torch.empty(K, N, device="cuda", dtype=BF))
    tb, n, tiles = ops.prep_table([e], "cuda"); ops.prep_weights(tb, n, tiles); return e
x = rnd(R, 256).to(BF); g, bt = 1 + 0.1 * rnd(256), 0.1 * rnd(256)
xhat, rstd = torch.empty(R, 256, device="cuda", dtype=BF), torch.empty(R, device="cuda")
e = prep(rnd(768, 256, sc=1 / 16), g, bt, rnd(768)); y = torch.empty(R, 768, device="cuda", dtype=BF)
for _ in range(3): ops.rowgemm(x, e["Wp"], y, R, 768, 256, bias=e["bp"], ln=True, xhat=xhat, rstd=rstd, stream_out=True)
Wb, bias, res = rnd(256, 256, sc=1 / 16).to(BF), rnd(256), rnd(R, 256).to(BF); y2 = torch.empty(R, 256, device="cuda", dtype=BF)
for _ in range(3): ops.rowgemm(x, Wb, y2, R, 256, 256, bias=bias, residual=res, ldr=256)
for _ in range(3): ops.rowgemm(x, Wb, y2, R, 256, 256)
dy = rnd(R, 768).to(BF); e3 = prep(rnd(768, 256, sc=1 / 28), g); dx = torch.empty(R, 256, device="cuda", dtype=BF)
for _ in range(3): ops.rowgemm(dy, e3["WpT"], dx, R, 256, 768, ldw=768, residual=res, ldr=256, ln_bwd=True, bwd_xhat=xhat, bwd_rstd=rstd)
dy5 = rnd(R, 512).to(BF); e5 = prep(rnd(512, 256, sc=1 / 22), g)
for _ in range(3): ops.rowgemm(dy5, e5["WpT"], dx, R, 256, 512, ldw=512, residual=res, ldr=256, ln_bwd=True, bwd_xhat=xhat, bwd_rstd=rstd)
up, dn = prep(rnd(512, 256, sc=1 / 16), g, bt, 0.1 * rnd(512)), prep(rnd(256, 512, sc=1 / 22), None, None, 0.1 * rnd(256))
st = torch.zeros(2, dtype=torch.int32, device="cuda"); ops.rng_seed(st, 1); drop = ops.dropout(st, 3, 0.4)
d = ops.mlp_desc(R, x=x, w_up=up["Wp"], b_up=up["bp"], w_down=dn["WpP"], b_down=dn["bp"], y=y2, xhat=xhat, rstd=rstd, drop=drop)
for _ in range(3): ops.mlp_fwd(d)
t1, gg, du = torch.empty(R, 256, device="cuda", dtype=BF), torch.empty(R, 512, device="cuda", dtype=BF), torch.empty(R, 512, device="cuda", dtype=BF)
d2 = ops.mlp_desc(R, w_up=up["Wp"], b_up=up["bp"], drop=drop, xhat=xhat, rstd=rstd, dy=res, w_down_t=dn["WpT"], w_up_t=up["WpTP"], t1=t1, g=gg, du=du, dx=dx)
for _ in range(3): ops.mlp_bwd(d2)
d3 = ops.mlp_desc(R, w_up=up["Wp"], b_up=up["bp"], drop=drop, xhat=xhat, dy=res, w_down_t=dn["WpT"], t1=t1, g=gg, du=du, dx=None)      # front half (round 4 default)
for _ in range(3): ops.mlp_bwd(d3)
torch.cuda.synchronize()

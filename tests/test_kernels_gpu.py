"""Per-kernel parity: every C-ABI entry point against a plain torch fp32 reference of the same op
(the oracle's functions where one exists).  Runs on the MI355X only."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import mm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from multi_modal_foundation_model_amd import _lib as L, ops as K
    L.check(L.lib().mmfm_device_check(0), "device_check")
    return K


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def close(a, b, rtol=2e-5, atol=2e-5, msg=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{msg}: max abs err {err:.3e} (ref max {ref:.3e})"


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(200, 256, 256), (333, 668, 1336), (3200, 768, 256), (300, 2, 256), (100, 4, 2), (64, 256, 4), (1600, 1336, 668)])
def test_gemm_linear_forward(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = torch.empty(M, N, device="cuda")
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b)
    close(y, x.double() @ w.double().T + b.double(), msg=f"linear {M}x{N}x{K}")


def test_gemm_epilogues(ops):
    M, N, K = 500, 512, 256
    x, w, b, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    y, pre = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, pre_out=pre, act=1)
    u = x @ w.T + b
    close(pre, u, msg="pre_out")
    close(y, torch.nn.functional.gelu(u), msg="gelu")
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2, act_scale=1.5)
    close(y, O.softsign(u) * 1.5, msg="softsign")
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, residual=res, ldr=N)
    close(y, u + res, msg="residual")


@pytest.mark.parametrize("M,N,K", [(700, 256, 512), (450, 668, 256), (200, 1336, 256), (128, 2, 4)])
def test_gemm_dx_and_actgrad(ops, M, N, K):
    """dX = dY[M,K] @ W[K,N]  (W is an nn.Linear weight [out=K, in=N]); optional act' multiply."""
    dy, w, pre = rnd(M, K, seed=5), rnd(K, N, seed=6, scale=K ** -0.5), rnd(M, N, seed=7)
    dx = torch.empty(M, N, device="cuda")
    ops.gemm(dy, w, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0)
    close(dx, dy.double() @ w.double(), msg="dX")
    ops.gemm(dy, w, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0, act=3, gradmul_pre=pre)
    pr = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(pr).sum().backward()
    close(dx, (dy @ w) * pr.grad, msg="dX*gelu'")
    ops.gemm(dy, w, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0, act=4, act_scale=2.0, gradmul_pre=pre)
    close(dx, (dy @ w) * 2.0 / (1 + pre.abs()) ** 2, msg="dX*softsign'")


@pytest.mark.parametrize("R,N,K,splits", [(3200, 256, 256, 8), (1000, 668, 256, 4), (777, 1336, 668, 3), (640, 2, 256, 2), (3200, 256, 512, 1)])
def test_gemm_dw_splitk(ops, R, N, K, splits):
    """dW[N,K] = dY[R,N]^T @ X[R,K] with split-K slabs + deterministic reduce."""
    dy, x = rnd(R, N, seed=8), rnd(R, K, seed=9)
    ref = dy.double().T @ x.double()
    dw = torch.empty(N, K, device="cuda")
    if splits == 1:
        ops.gemm(dy, x, dw, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0)
    else:
        kchunk = ((R + splits - 1) // splits + 63) // 64 * 64
        splits = (R + kchunk - 1) // kchunk
        slabs = torch.full((splits, N, K), float("nan"), device="cuda")
        ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=splits, kchunk=kchunk, slab_stride=N * K)
        ops.reduce_slabs(dw, slabs, N * K, splits, N * K)
    close(dw, ref, rtol=1e-4, atol=1e-3 * R ** 0.5 / 30, msg="dW")


def test_colsum(ops):
    for R, N in [(3200, 256), (1001, 668), (64, 2), (51200, 768), (333, 1336)]:
        x = rnd(R, N, seed=10)
        out = torch.empty(N, device="cuda")
        ws = torch.empty(256 * N + 16, device="cuda")
        ops.colsum(x, R, N, N, out, ws)
        close(out, x.double().sum(0), atol=1e-3, msg="colsum")
        ops.colsum(x, R, N, N, out, ws, accumulate=True)
        close(out, 2 * x.double().sum(0), atol=2e-3, msg="colsum acc")


def test_gemm_argument_errors(ops):
    from multi_modal_foundation_model_amd._lib import MmfmError
    x = rnd(8, 8)
    with pytest.raises(MmfmError):
        ops.gemm(x, x, x, 8, 8, 8, lda=4, ldb=8, ldc=8)
    with pytest.raises(MmfmError):
        ops.gemm(x, x, x, 8, 8, 8, lda=8, ldb=8, ldc=8, act=3)


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("R,H", [(3200, 256), (33, 32), (77, 512), (5, 1024)])
def test_layernorm(ops, R, H):
    x, g, b, dy, dres = rnd(R, H, seed=1, scale=2.0), rnd(H, seed=2), rnd(H, seed=3), rnd(R, H, seed=4), rnd(R, H, seed=5)
    y, mean, rstd = torch.empty_like(x), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    ops.layernorm_fwd(x, g, b, y, mean, rstd, R, H)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-5)
    close(y, yr, msg="ln fwd")
    close(mean, x.mean(-1), msg="mean")
    yr.backward(dy)
    dx, dg, db = torch.empty_like(x), torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    ws = torch.empty(1024 * 2 * H, device="cuda")
    ops.layernorm_bwd(dy, x, mean, rstd, g, dres, dx, dg, db, R, H, ws)
    close(dx, xr.grad + dres, rtol=1e-4, atol=1e-4, msg="ln dx")
    close(dg, gr.grad, rtol=1e-4, atol=1e-3, msg="ln dgamma")
    close(db, br.grad, rtol=1e-4, atol=1e-3, msg="ln dbeta")
    ops.layernorm_bwd(dy, x, mean, rstd, g, None, dx, dg, db, R, H, ws, accumulate=True)
    close(dx, xr.grad, rtol=1e-4, atol=1e-4, msg="ln dx (no dres)")
    close(dg, 2 * gr.grad, rtol=1e-4, atol=2e-3, msg="ln dgamma acc")


def test_layernorm_destitch(ops):
    B, M, T, H = 3, 2, 5, 32
    L = M * T
    x, g, b = rnd(B * L, H, seed=1), rnd(H, seed=2), rnd(H, seed=3)
    y, mean, rstd = torch.empty_like(x), torch.empty(B * L, device="cuda"), torch.empty(B * L, device="cuda")
    ops.layernorm_fwd(x, g, b, y, mean, rstd, B * L, H, ds_L=L, ds_T=T)
    ref = torch.nn.functional.layer_norm(x, (H,), g, b, 1e-5).view(B, M, T, H).permute(1, 0, 2, 3).reshape(B * L, H)
    close(y, ref, msg="destitched ln")
    dy = rnd(B * L, H, seed=4)          # laid out [M][B*T][H]
    dx, dg, db = torch.empty_like(x), torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    ws = torch.empty(1024 * 2 * H, device="cuda")
    ops.layernorm_bwd(dy, x, mean, rstd, g, None, dx, dg, db, B * L, H, ws, ds_L=L, ds_T=T)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (H,), g, b, 1e-5).backward(dy.view(M, B, T, H).permute(1, 0, 2, 3).reshape(B * L, H))
    close(dx, xr.grad, rtol=1e-4, atol=1e-4, msg="destitched ln bwd")


# ------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, mask, scale):
    s = (q @ k.transpose(-1, -2)) * scale
    s = s.masked_fill(~mask[:, None], float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,heads,L,dh,flags", [(2, 8, 200, 32, 1), (2, 8, 200, 32, 0), (3, 4, 16, 8, 1), (2, 4, 16, 8, 2),
                                                (2, 4, 16, 8, 4), (2, 2, 70, 16, 1), (1, 8, 100, 64, 1),
                                                # tiled kernels (operands do not fit LDS): BASELINE config 5 is L=600, dh=64
                                                (2, 8, 600, 64, 1), (2, 2, 600, 64, 0), (1, 2, 330, 32, 4), (1, 4, 620, 16, 2)])
def test_attention_fwd_bwd(ops, B, heads, L, dh, flags):
    from multi_modal_foundation_model_amd import _lib as Lb
    H = heads * dh
    qkv = rnd(B * L, 3 * H, seed=1)
    d_o = rnd(B * L, H, seed=2)
    keypad = torch.ones(B, L, dtype=torch.uint8)
    keypad[0, L - 3:] = 0
    if B > 1:
        keypad[1, L // 2: L // 2 + 2] = 0
    mod_id = (torch.arange(L) >= L // 2).to(torch.uint8)
    kp, mi = keypad.cuda(), mod_id.cuda()
    o, lse = torch.empty(B * L, H, device="cuda"), torch.empty(B, heads, L, device="cuda")
    dqkv = torch.full((B * L, 3 * H), float("nan"), device="cuda")
    scale = 1.0 / math.sqrt(dh)
    es = 4
    base = qkv.data_ptr()
    desc = ops.attn_desc(Lb.F32, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, mi,
                         flags, scale, d_o=d_o.data_ptr(), lddo=H, dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + H * es,
                         dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    # reference
    kpb = kp.bool()
    if flags & 2:
        m = torch.tril(torch.ones(L, L, dtype=torch.bool, device="cuda"))[None].expand(B, L, L)
    else:
        m = kpb[:, None, :].expand(B, L, L)
    if flags & 1:
        m = m | torch.eye(L, dtype=torch.bool, device="cuda")[None]
    if flags & 4:
        m = m | (mi[None, :, None] != mi[None, None, :])
    x = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, L, heads, dh).transpose(1, 2) for t in x.split(H, dim=1)]
    oref = ref_attention(q, k, v, m, scale).transpose(1, 2).reshape(B * L, H)
    close(o, oref, rtol=1e-4, atol=2e-5, msg="attn fwd")
    s = (q @ k.transpose(-1, -2)) * scale
    lref = torch.logsumexp(s.masked_fill(~m[:, None], float("-inf")), -1)
    close(lse, lref, rtol=1e-5, atol=1e-4, msg="lse")
    oref.backward(d_o)
    close(dqkv, x.grad, rtol=1e-3, atol=5e-5, msg="attn bwd")


@pytest.mark.parametrize("Lq,Lk,dh", [(40, 72, 32), (40, 700, 64), (300, 70, 64)])
def test_attention_cross_shapes(ops, Lq, Lk, dh):
    """Lq != Lk (no DIAG): cross attention over a longer context (the long cases run the tiled kernels)."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads = 2, 4
    H = heads * dh
    q, kv, d_o = rnd(B * Lq, H, seed=1), rnd(B * Lk, 2 * H, seed=2), rnd(B * Lq, H, seed=3)
    kp = torch.ones(B, Lk, dtype=torch.uint8)
    kp[1, Lk - 12:] = 0
    kp = kp.cuda()
    o, lse = torch.empty(B * Lq, H, device="cuda"), torch.empty(B, heads, Lq, device="cuda")
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    desc = ops.attn_desc(Lb.F32, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 4, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, 0, dh ** -0.5, d_o=d_o.data_ptr(), lddo=H, dq=dq.data_ptr(), dk=dkv.data_ptr(),
                         dv=dkv.data_ptr() + H * 4, lddq=H, lddk=2 * H, lddv=2 * H)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    qr, kvr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    Q = qr.view(B, Lq, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, Lk, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    oref = ref_attention(Q, K_, V_, kp.bool()[:, None, :].expand(B, Lq, Lk), dh ** -0.5).transpose(1, 2).reshape(B * Lq, H)
    close(o, oref, rtol=1e-4, atol=2e-5, msg="xattn fwd")
    oref.backward(d_o)
    close(dq, qr.grad, rtol=1e-3, atol=5e-5, msg="xattn dq")
    close(dkv, kvr.grad, rtol=1e-3, atol=5e-5, msg="xattn dkv")


@pytest.mark.parametrize("L,dh", [(64, 32), (600, 64)])
def test_attention_dropout_consistency(ops, L, dh):
    """Dropout masks are regenerated in the backward: check dV against the forward's own mask by
    linearity: with q=k=0 all probabilities are uniform, so O = mean over kept keys of V/(1-p).
    (600, 64) runs the tiled kernels."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads, p = 2, 4, 0.4
    H = heads * dh
    qkv = torch.zeros(B * L, 3 * H, device="cuda")
    qkv[:, 2 * H:] = 1.0                                    # V = 1  => O[q] = kept_fraction/(1-p)
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 1234)
    kp = torch.ones(B, L, dtype=torch.uint8, device="cuda")
    o, lse = torch.empty(B * L, H, device="cuda"), torch.empty(B, heads, L, device="cuda")
    d_o = torch.ones(B * L, H, device="cuda")
    dqkv = torch.empty(B * L, 3 * H, device="cuda")
    base, es = qkv.data_ptr(), 4
    desc = ops.attn_desc(Lb.F32, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None, 0,
                         dh ** -0.5, drop_p=ops.dropout(state, 7, p), d_o=d_o.data_ptr(), lddo=H, dq=dqkv.data_ptr(),
                         dk=dqkv.data_ptr() + H * es, dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
    ops.attn_fwd(desc)
    o1 = o.clone()
    ops.attn_fwd(desc)
    assert torch.equal(o, o1), "same state/site must give the same mask"
    frac = o.view(B, L, heads, dh)[..., 0] * (1 - p)            # kept fraction per (b,q,h)
    assert abs(frac.mean().item() - (1 - p)) < 0.02
    assert frac.std().item() > 0.01
    ops.attn_bwd(desc)
    dv = dqkv[:, 2 * H:].view(B, L, heads, dh)[..., 0]          # dV[k] = sum_q keep(q,k)/((1-p) L)
    # sum_k dV[k] == sum_q O[q]  (both count kept (q,k) pairs)
    close(dv.sum(1), o.view(B, L, heads, dh)[..., 0].sum(1), rtol=1e-4, atol=1e-3, msg="fwd/bwd dropout mask agree")
    ops.rng_advance(state)
    ops.attn_fwd(desc)
    assert not torch.equal(o, o1), "advancing the RNG state must change the mask"


# ------------------------------------------------------------------------------------ masks / stitch / loss
def test_mask_prep_bit_exact_vs_reference_fixture(ops):
    z, cases = load_npz("mask_index_ops.npz")
    attn = torch.from_numpy(z["attn"])
    B, T = attn.shape
    for key in cases:
        ms = [torch.from_numpy(z[f"{key}/in_mask/{m}"]) for m in ("ap", "behavior")]
        # feed un-anded masks with a channel dim to exercise stride + the '& attn' (mm.py:270)
        full = [m[:, :, None].repeat(1, 1, 3).contiguous().cuda() for m in ms]
        L = 2 * T
        tok, kpd = torch.empty(B, L, dtype=torch.uint8, device="cuda"), torch.empty(B, L, dtype=torch.uint8, device="cuda")
        keep0, mod = torch.empty(L, dtype=torch.uint8, device="cuda"), torch.empty(L, dtype=torch.uint8, device="cuda")
        cnt = torch.empty(2, dtype=torch.int64, device="cuda")
        ops.mask_prep(B, T, full, [3, 3], attn.cuda(), [5, 2], tok, kpd, keep0, mod, cnt)
        enc_mask = torch.from_numpy(z[f"{key}/enc_mask"])
        np.testing.assert_array_equal(tok.cpu().numpy(), enc_mask.numpy().astype(np.uint8))
        np.testing.assert_array_equal(kpd.cpu().numpy(), torch.cat([attn, attn], 1).numpy().astype(np.uint8))
        np.testing.assert_array_equal(mod.cpu().numpy(), z[f"{key}/enc_mod_mask"][0].astype(np.uint8))
        np.testing.assert_array_equal(keep0.cpu().numpy(), (enc_mask[0] != 1).numpy().astype(np.uint8))
        assert cnt.tolist() == [int(enc_mask[:, :T].sum()) * 5, int(enc_mask[:, T:].sum()) * 2]
        # zeroed-token positions == the reference's (sample-0 quirk)
        xs = [torch.from_numpy(z[f"{key}/in_x/{m}"]) for m in ("ap", "behavior")]
        toks = torch.cat(xs, 1) * keep0.cpu()[None, :, None]
        np.testing.assert_array_equal(toks.numpy(), z[f"{key}/enc_tokens"])


def test_stitch_fwd_bwd(ops):
    B, T, M, H, max_F = 5, 7, 2, 32, 9
    L = M * T
    g = torch.Generator().manual_seed(3)
    ts = torch.randint(0, max_F, (B, T), generator=g).cuda()
    keep0 = (torch.rand(L, generator=g) > 0.3).to(torch.uint8).cuda()
    x, emb = torch.zeros(B, L, H, device="cuda"), torch.zeros(B, L, H, device="cuda")
    toks = [rnd(B * T, H, seed=10 + m).requires_grad_(True) for m in range(M)]
    mods = [rnd(H, seed=20 + m).requires_grad_(True) for m in range(M)]
    poss = [rnd(max_F, H, seed=30 + m).requires_grad_(True) for m in range(M)]
    for m in range(M):
        ops.stitch_fwd(toks[m].detach(), mods[m].detach(), poss[m].detach(), ts, keep0, x, emb, B, T, L, m, H, max_F)
    e_ref = torch.cat([mods[m][None, None, :] + poss[m][ts] for m in range(M)], 1)
    x_ref = torch.cat([t.view(B, T, H) for t in toks], 1) * keep0[None, :, None] + e_ref
    close(emb, e_ref, msg="emb")
    close(x, x_ref, msg="x")
    dx, dextra = rnd(B, L, H, seed=40), rnd(B, L, H, seed=41)
    (x_ref * dx + e_ref * dextra).sum().backward()
    ws = torch.empty(64 * (max_F + 1) * H * 8, device="cuda")
    for m in range(M):
        d_tok, d_mod, d_pos = torch.empty(B * T, H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(max_F, H, device="cuda")
        ops.stitch_bwd(dx, dextra, ts, keep0, None, d_tok, d_mod, d_pos, False, False, B, T, L, m, H, max_F, ws)
        close(d_tok, toks[m].grad, msg="d_tok")
        close(d_mod, mods[m].grad, atol=1e-4, msg="d_mod")
        close(d_pos, poss[m].grad, atol=1e-4, msg="d_pos")
        ops.stitch_bwd(dx, dextra, ts, keep0, None, d_tok, d_mod, d_pos, True, False, B, T, L, m, H, max_F, ws)
        close(d_mod, 2 * mods[m].grad, atol=2e-4, msg="d_mod accumulated")
        close(d_pos, poss[m].grad, atol=1e-4, msg="d_pos overwritten")


@pytest.mark.parametrize("B,T,M,H,max_F,with_drop", [(5, 7, 2, 32, 9, False), (40, 100, 2, 256, 100, True), (3, 200, 3, 512, 200, False)])
def test_stitch_bwd_bf16_onehot_gemm(ops, B, T, M, H, max_F, with_drop):
    """bf16 mode: the position/modality-embedding gradients are a one-hot matrix product on the MFMA GEMM.  fp32
    accumulation of bf16 values -> compare against an fp64 scatter of the same bf16 inputs."""
    from multi_modal_foundation_model_amd import _lib as Lb
    L = M * T
    g = torch.Generator().manual_seed(3)
    ts = torch.randint(0, max_F, (B, T), generator=g).cuda()
    keep0 = (torch.rand(L, generator=g) > 0.3).to(torch.uint8).cuda()
    dx, dextra = bf(rnd(B, L, H, seed=40)), bf(rnd(B, L, H, seed=41))
    ws = torch.empty(Lb.lib().mmfm_stitch_bwd_workspace(Lb.BF16, B, T, L, H, max_F), dtype=torch.uint8, device="cuda")
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 11)
    for m in range(M):
        for extra in (dextra, None):
            d_tok = torch.full((B * T, H), float("nan"), device="cuda", dtype=torch.bfloat16)
            d_mod, d_pos = torch.empty(H, device="cuda"), torch.empty(max_F, H, device="cuda")
            drop = ops.dropout(state, 5, 0.2) if with_drop else None
            ops.stitch_bwd(dx, extra, ts, keep0, drop, d_tok, d_mod, d_pos, False, False, B, T, L, m, H, max_F, ws)
            e = dx[:, m * T:(m + 1) * T].double() + (extra[:, m * T:(m + 1) * T].double() if extra is not None else 0)
            ref_pos = torch.zeros(max_F, H, dtype=torch.float64, device="cuda").index_add_(0, ts.reshape(-1), e.reshape(B * T, H))
            close(d_pos, ref_pos.float(), rtol=1e-5, atol=1e-4 * math.sqrt(B), msg="d_pos (one-hot GEMM)")
            close(d_mod, e.sum((0, 1)).float(), rtol=1e-5, atol=1e-4 * math.sqrt(B * T), msg="d_mod (one-hot GEMM)")
            tok_ref = dx[:, m * T:(m + 1) * T].reshape(B * T, H).float() * keep0[m * T:(m + 1) * T].repeat(B)[:, None]
            if with_drop:
                kept = d_tok.float() != 0
                close(d_tok.float()[kept], (tok_ref / 0.8)[kept], rtol=1e-2, atol=1e-3, msg="d_tok kept values")
                frac = kept.float().sum() / (tok_ref != 0).float().sum()
                assert abs(frac.item() - 0.8) < 0.02
            else:
                assert torch.equal(d_tok.float(), tok_ref)
            ops.stitch_bwd(dx, extra, ts, keep0, drop, d_tok, d_mod, d_pos, True, False, B, T, L, m, H, max_F, ws)
            close(d_mod, 2 * e.sum((0, 1)).float(), rtol=1e-5, atol=2e-4 * math.sqrt(B * T), msg="d_mod accumulated")


@pytest.mark.parametrize("kind,N", [(0, 668), (1, 2), (0, 12)])
def test_masked_loss(ops, kind, N):
    B, T = 6, 10
    R = B * T
    pred = rnd(R, N, seed=1, scale=0.5)
    tgt = torch.poisson(torch.full((R, N), 0.3)).cuda() if kind == 0 else rnd(R, N, seed=2)
    M = 2
    tokmask = (torch.rand(B, M * T) < 0.4).to(torch.uint8).cuda()
    rowmask = tokmask[:, T:]                                  # modality 1's slice: mask_ld = M*T
    out, ws = torch.empty(1, device="cuda"), torch.empty(1024, device="cuda")
    ops.masked_loss_fwd(kind, pred, tgt, rowmask, M * T, T, R, N, out, ws)
    pr = pred.clone().requires_grad_(True)
    el = (torch.exp(pr) - tgt * pr) if kind == 0 else (pr - tgt) ** 2
    mk = rowmask.reshape(R, 1).float()
    ref = (el * mk).sum()
    close(out[0], ref, rtol=1e-5, atol=1e-3, msg="loss sum")
    cnt = torch.tensor([int(rowmask.sum()) * N, 7], dtype=torch.int64, device="cuda")
    sums = torch.stack([out[0], torch.tensor(3.0, device="cuda")])
    loss, inv_n = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    ops.loss_finalize(sums, cnt, 2, loss, inv_n)
    n = int(cnt.sum())
    close(loss[0], (ref + 3.0) / n, rtol=1e-5, msg="loss")
    gout = torch.tensor([0.5], device="cuda")
    dpred = torch.empty_like(pred)
    ops.masked_loss_bwd(kind, pred, tgt, rowmask, M * T, T, R, N, gout, inv_n, dpred)
    (0.5 * ref / n).backward()
    close(dpred, pr.grad, rtol=1e-5, atol=1e-8, msg="dpred")
    # nothing masked -> 0/0 = NaN like the reference (mm.py:237)
    ops.loss_finalize(torch.zeros(2, device="cuda"), torch.zeros(2, dtype=torch.int64, device="cuda"), 2, loss, inv_n)
    assert torch.isnan(loss[0])


# ------------------------------------------------------------------------------------ optimiser / dropout
def test_adamw_matches_torch(ops):
    n = 10007
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.1)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-4, weight_decay=0.01, eps=1e-8)
    for step in range(1, 4):
        lr, b1 = O.onecycle(step - 1, 100)
        opt.param_groups[0]["lr"], opt.param_groups[0]["betas"] = lr, (b1, 0.999)
        ref.grad = g.clone()
        opt.step()
        bc1, bc2 = 1 - b1 ** step, 1 - 0.999 ** step
        hyper = torch.tensor([1 - lr * 0.01, 1 - b1, 0.999, 1 - 0.999, lr / bc1, math.sqrt(bc2), 1e-8, 1.0], device="cuda")
        ops.adamw_step(p, g, m, v, None, n, hyper)
        close(p, ref.data, rtol=1e-6, atol=1e-7, msg=f"adamw step {step}")


def test_gemm_dropout_matches_dropout_apply(ops):
    M, N, K, p = 300, 256, 64, 0.4
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 99)
    y0, y1, y2 = (torch.empty(M, N, device="cuda") for _ in range(3))
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N)
    ops.gemm(x, w, y1, M, N, K, lda=K, ldb=K, ldc=N, drop=ops.dropout(state, 5, p))
    ops.dropout_apply(y0, y2, M, N, ops.dropout(state, 5, p))
    assert torch.equal(y1, y2), "GEMM-epilogue dropout and dropout_apply must share the mask"
    kept = (y1 != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.01
    nz = y1 != 0
    close(y1[nz], y0[nz] / (1 - p), msg="scaled by 1/(1-p)")
    ops.dropout_apply(y0, y2, M, N, ops.dropout(state, 6, p))
    assert not torch.equal(y1, y2), "different site -> different mask"


# ------------------------------------------------------------------------------------ bf16 MFMA GEMM
def bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N", [(300, 256), (37, 12)])         # 16-B vector kernel / scalar kernel of mmfm_dropout_apply
def test_gemm_bf16_dropout_matches_dropout_apply(ops, M, N):
    """The backward of an MLP-output dropout re-applies the forward's mask with mmfm_dropout_apply: the mask must be the one
    the bf16 GEMM epilogue drew (same site, same counter m*N + n), and kept values scale by 1/(1-p)."""
    K, p = 64, 0.4
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2))
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 99)
    y0, y1, y2 = (torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N)
    ops.gemm(x, w, y1, M, N, K, lda=K, ldb=K, ldc=N, drop=ops.dropout(state, 5, p))
    ops.dropout_apply(y0, y2, M, N, ops.dropout(state, 5, p))
    assert torch.equal((y1 != 0), (y2 != 0)) or ((y1 != 0) ^ (y2 != 0)).float().mean().item() < 1e-3     # exact zeros in y0 aside
    nz = y2 != 0
    close_bf16(y2[nz], (y0.float() / (1 - p))[nz], "bf16 dropout_apply scale", tol=1e-2)
    assert abs(nz.float().mean().item() - (1 - p)) < (0.02 if M * N > 10000 else 0.15)


def close_bf16(a, b, msg, tol=1.5e-2):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-6
    assert err <= tol * scale, f"{msg}: max abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("M,N,K", [(200, 256, 256), (333, 668, 1336), (3200, 768, 256), (1600, 1336, 668), (300, 2, 256), (100, 4, 2), (64, 256, 4)])
def test_gemm_bf16_linear_forward(ops, M, N, K):
    x, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b)
    close_bf16(y, x.double() @ w.double().T + b.double(), f"bf16 linear {M}x{N}x{K}")


@pytest.mark.parametrize("N", [512, 668, 12])          # 668 = 83 x 8 + 4: the 4-column-granular vector epilogue; 12: ragged, single tile
def test_gemm_bf16_epilogues_and_layouts(ops, N):
    M, K = 500, 256
    x, w, b, res = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3), bf(rnd(M, N, seed=4))
    y, pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, pre_out=pre, act=1, residual=res, ldr=N)
    u = x.float() @ w.float().T + b
    close_bf16(pre, u, "bf16 pre_out")
    close_bf16(y, torch.nn.functional.gelu(u) + res.float(), "bf16 gelu+residual")
    # dX = dY @ W  (W [K,N] row-contiguous -> transposed LDS reads) with gelu' multiply
    dy, w2, pr = bf(rnd(M, K, seed=5)), bf(rnd(K, N, seed=6, scale=K ** -0.5)), bf(rnd(M, N, seed=7))
    dx = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w2, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0)
    close_bf16(dx, dy.float() @ w2.float(), "bf16 dX")
    ops.gemm(dy, w2, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0, act=3, gradmul_pre=pr)
    p32 = pr.float().requires_grad_(True)
    torch.nn.functional.gelu(p32).sum().backward()
    close_bf16(dx, (dy.float() @ w2.float()) * p32.grad, "bf16 dX*gelu'")
    # act 4 / act 5: softsign' from the saved pre-activation, and from the activation's output (what the bf16 engine keeps)
    ops.gemm(dy, w2, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0, act=4, act_scale=0.7, gradmul_pre=pr)
    exact = (dy.double() @ w2.double()) * 0.7 / (1 + pr.double().abs()) ** 2
    close_bf16(dx, exact, "bf16 dX*softsign'")
    a_out = bf(0.7 * pr.double() / (1 + pr.double().abs()))
    ops.gemm(dy, w2, dx, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0, act=5, act_scale=0.7, gradmul_pre=a_out)
    close_bf16(dx, (dy.double() @ w2.double()) * 0.7 * (1 - a_out.double().abs() / 0.7) ** 2, "bf16 dX*softsign' from the output")
    close_bf16(dx, exact, "bf16 dX*softsign' from the output vs exact", tol=2.5e-2)


@pytest.mark.parametrize("R,N,K,splits", [(3200, 256, 256, 8), (1000, 668, 256, 4), (777, 1336, 668, 3), (640, 2, 256, 2), (3200, 768, 256, 1), (900, 4, 2, 2)])
def test_gemm_bf16_dw_splitk(ops, R, N, K, splits):
    dy, x = bf(rnd(R, N, seed=8)), bf(rnd(R, K, seed=9))
    ref = dy.double().T @ x.double()
    dw = torch.empty(N, K, device="cuda")
    if splits == 1:
        ops.gemm(dy, x, dw, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, c_f32=1)
    else:
        kchunk = ((R + splits - 1) // splits + 63) // 64 * 64
        splits = (R + kchunk - 1) // kchunk
        slabs = torch.full((splits, N, K), float("nan"), device="cuda")
        ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=splits, kchunk=kchunk, slab_stride=N * K, c_f32=1)
        ops.reduce_slabs(dw, slabs, N * K, splits, N * K)
    close_bf16(dw, ref, "bf16 dW", tol=2e-3)          # fp32 accumulate of exact bf16 products


@pytest.mark.parametrize("R,N,K,splits", [(3200, 256, 256, 8), (1000, 668, 256, 4), (777, 1336, 668, 3), (640, 2, 256, 2), (3200, 768, 256, 1),
                                          (900, 4, 2, 2), (51200, 512, 256, 40),
                                          # the streaming dW kernel (csrc/gemm_dw.hip): ragged tile edges, a reduction that is not a multiple
                                          # of the k-tile, a short last slab, one item per workgroup at the bench size
                                          (1000, 200, 72, 3), (1000, 200, 72, 1), (8250, 264, 520, 7), (204800, 768, 256, 42), (204800, 256, 512, 64), (51200, 256, 128, 30)])
def test_gemm_bf16_dw_with_fused_bias_grad(ops, R, N, K, splits):
    """mmfm_gemm_desc.colsum: the bias gradient (column sums of dY) computed by the dW launch, its partials stored
    behind each weight slab so that one slab reduction yields [dW | db] (the flat gradient buffer's layout)."""
    dy, x = bf(rnd(R, N, seed=8)), bf(rnd(R, K, seed=9))
    ref_w, ref_b = dy.double().T @ x.double(), dy.double().sum(0)
    out = torch.full((N * K + N,), float("nan"), device="cuda")
    if splits == 1:
        ops.gemm(dy, x, out, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, c_f32=1, colsum=out.data_ptr() + 4 * N * K)
    else:
        kchunk = ((R + splits - 1) // splits + 63) // 64 * 64
        splits = (R + kchunk - 1) // kchunk
        stride = (N * K + N + 7) // 8 * 8
        slabs = torch.full((splits, stride), float("nan"), device="cuda")
        ops.gemm(dy, x, slabs, N, K, R, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=splits, kchunk=kchunk, slab_stride=stride,
                 c_f32=1, colsum=slabs.data_ptr() + 4 * N * K)
        ops.reduce_slabs(out, slabs, N * K + N, splits, stride)
    close_bf16(out[:N * K].view(N, K), ref_w, "bf16 dW (fused)", tol=2e-3)
    close(out[N * K:], ref_b.float(), rtol=1e-5, atol=1e-4 * math.sqrt(R), msg="fused bias grad")


@pytest.mark.parametrize("B,heads,L,dh,flags", [(2, 8, 200, 32, 1), (2, 8, 200, 32, 0), (2, 4, 48, 16, 2), (2, 4, 40, 16, 4), (2, 2, 70, 64, 1),
                                                (3, 4, 16, 8, 1), (2, 8, 600, 64, 1), (2, 2, 460, 64, 0),
                                                # tiled bf16 kernels at the other head dims (images beyond 160 KB of LDS)
                                                (1, 2, 1100, 32, 1), (1, 2, 1100, 32, 4), (1, 1, 1700, 16, 2)])
def test_attention_bf16_fwd_bwd(ops, B, heads, L, dh, flags):
    """bf16 MFMA attention (dh 16/32/64; dh=8 falls through to fp32 compute on bf16 storage) vs an fp32
    reference on the same bf16-rounded inputs.  Tolerance: bf16 has 8 significant bits."""
    from multi_modal_foundation_model_amd import _lib as Lb
    H = heads * dh
    qkv = bf(rnd(B * L, 3 * H, seed=1))
    d_o = bf(rnd(B * L, H, seed=2))
    keypad = torch.ones(B, L, dtype=torch.uint8)
    keypad[0, L - 3:] = 0
    if B > 1:
        keypad[1, L // 2: L // 2 + 2] = 0
    mod_id = (torch.arange(L) >= L // 2).to(torch.uint8)
    kp, mi = keypad.cuda(), mod_id.cuda()
    o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
    dqkv = torch.full((B * L, 3 * H), float("nan"), device="cuda", dtype=torch.bfloat16)
    scale, es, base = 1.0 / math.sqrt(dh), 2, qkv.data_ptr()
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, mi,
                         flags, scale, d_o=d_o.data_ptr(), lddo=H, dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + H * es,
                         dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    kpb = kp.bool()
    m = torch.tril(torch.ones(L, L, dtype=torch.bool, device="cuda"))[None].expand(B, L, L) if flags & 2 else kpb[:, None, :].expand(B, L, L)
    if flags & 1:
        m = m | torch.eye(L, dtype=torch.bool, device="cuda")[None]
    if flags & 4:
        m = m | (mi[None, :, None] != mi[None, None, :])
    x = qkv.float().requires_grad_(True)
    q, k, v = [t.view(B, L, heads, dh).transpose(1, 2) for t in x.split(H, dim=1)]
    oref = ref_attention(q, k, v, m, scale).transpose(1, 2).reshape(B * L, H)
    close_bf16(o, oref, "bf16 attn fwd", tol=2e-2)
    s = (q @ k.transpose(-1, -2)) * scale
    close(lse, torch.logsumexp(s.masked_fill(~m[:, None], float("-inf")), -1), rtol=1e-3, atol=2e-3, msg="bf16 lse")
    oref.backward(d_o.float())
    for nm, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        close_bf16(dqkv[:, sl], x.grad[:, sl], f"bf16 attn {nm}", tol=3e-2)


@pytest.mark.parametrize("Lq,Lk,dh", [(72, 40, 32), (40, 72, 32), (200, 200, 64), (224, 100, 16)])
def test_attention_bf16_cross_shapes(ops, Lq, Lk, dh):
    """bf16 attention with Lq != Lk and no DIAG: Lk <= Lq runs the single-pass backward, Lk > Lq the two-phase kernels."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads = 2, 4
    H = heads * dh
    q, kv, d_o = bf(rnd(B * Lq, H, seed=1)), bf(rnd(B * Lk, 2 * H, seed=2)), bf(rnd(B * Lq, H, seed=3))
    kp = torch.ones(B, Lk, dtype=torch.uint8)
    kp[1, Lk - 5:] = 0
    kp = kp.cuda()
    o, lse = torch.empty(B * Lq, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, Lq, device="cuda")
    dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
    desc = ops.attn_desc(Lb.BF16, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, 0, dh ** -0.5, d_o=d_o.data_ptr(), lddo=H, dq=dq.data_ptr(), dk=dkv.data_ptr(),
                         dv=dkv.data_ptr() + H * 2, lddq=H, lddk=2 * H, lddv=2 * H)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    Q = qr.view(B, Lq, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, Lk, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    oref = ref_attention(Q, K_, V_, kp.bool()[:, None, :].expand(B, Lq, Lk), dh ** -0.5).transpose(1, 2).reshape(B * Lq, H)
    close_bf16(o, oref, "bf16 xattn fwd", tol=2e-2)
    oref.backward(d_o.float())
    close_bf16(dq, qr.grad, "bf16 xattn dq", tol=3e-2)
    close_bf16(dkv, kvr.grad, "bf16 xattn dkv", tol=3e-2)


@pytest.mark.parametrize("L,dh", [(64, 32), (600, 64), (460, 64)])
def test_attention_bf16_dropout_consistency(ops, L, dh):
    """(600, 64): neither bf16 kernel fits -> tiled fp32 compute; (460, 64): the bf16 forward alone would fit but the
    backward does not, so the pair must move together (one dropout hash layout per forward/backward pair)."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads, p = 2, 4, 0.4
    H = heads * dh
    qkv = torch.zeros(B * L, 3 * H, device="cuda", dtype=torch.bfloat16)
    qkv[:, 2 * H:] = 1.0
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 1234)
    kp = torch.ones(B, L, dtype=torch.uint8, device="cuda")
    o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
    d_o = torch.ones(B * L, H, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty(B * L, 3 * H, device="cuda", dtype=torch.bfloat16)
    base, es = qkv.data_ptr(), 2
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp, None, 0,
                         dh ** -0.5, drop_p=ops.dropout(state, 7, p), d_o=d_o.data_ptr(), lddo=H, dq=dqkv.data_ptr(),
                         dk=dqkv.data_ptr() + H * es, dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
    ops.attn_fwd(desc)
    frac = o.float().view(B, L, heads, dh)[..., 0] * (1 - p)
    assert abs(frac.mean().item() - (1 - p)) < 0.02 and frac.std().item() > 0.01
    ops.attn_bwd(desc)
    dv = dqkv[:, 2 * H:].float().view(B, L, heads, dh)[..., 0]
    close(dv.sum(1), o.float().view(B, L, heads, dh)[..., 0].sum(1), rtol=2e-2, atol=0.5, msg="bf16 fwd/bwd dropout mask agree")


@pytest.mark.parametrize("R,kchunk", [(6400, 1600), (40960, 2048)])
def test_gemm_bf16_dw_stream_padded_rows(ops, R, kchunk):
    """The tokeniser's weight gradient (1336 x 668, K = B*T tokens): X rows padded to a 16-B multiple (ld 672) so the streaming kernel takes the
    shape although N % 8 = 4; the pad columns hold garbage that must not reach the 668 real ones.  R = 6,400 runs the 128-wide tile
    instantiation, R = 40,960 (>= 32,768: the step's regime) the 256-wide one - gemm_dw_kernel<256, 32>, the only one in the B = 1024 step."""
    N, K, ld = 1336, 668, 672
    dy = bf(rnd(R, N, seed=18))
    xp = torch.full((R, ld), float("nan"), device="cuda", dtype=torch.bfloat16)
    x = bf(rnd(R, K, seed=19))
    xp[:, :K] = x
    splits = R // kchunk
    stride = (N * K + N + 7) // 8 * 8
    slabs = torch.full((splits, stride), float("nan"), device="cuda")
    out = torch.empty(N * K + N, device="cuda")
    ops.gemm(dy, xp, slabs, N, K, R, lda=N, ldb=ld, ldc=K, a_kcontig=0, b_kcontig=0, splits=splits, kchunk=kchunk, slab_stride=stride, c_f32=1,
             colsum=slabs.data_ptr() + 4 * N * K)
    ops.reduce_slabs(out, slabs, N * K + N, splits, stride)
    close_bf16(out[:N * K].view(N, K), dy.double().T @ x.double(), "padded-row dW", tol=2e-3)
    close(out[N * K:], dy.double().sum(0).float(), rtol=1e-5, atol=1e-4 * math.sqrt(R), msg="padded-row bias grad")


@pytest.mark.parametrize("M,N,K,Kreal", [(2048 + 100, 512, 512, 512), (1500, 1336, 704, 668), (1100, 256, 1344, 1336), (4096 + 37, 1536, 512, 512),
                                          (1024, 1024, 1024, 1024)])
def test_gemm_bf16_256_tile_kernel(ops, M, N, K, Kreal, monkeypatch):
    monkeypatch.setenv("MMFM_GEMM_BIG_MIN_TILES", "1")          # the launcher keeps problems of fewer than 96 tiles on the 128-tile kernel
    _gemm_bf16_256_tile_kernel(ops, M, N, K, Kreal)


def _gemm_bf16_256_tile_kernel(ops, M, N, K, Kreal):
    """csrc/gemm_big.hip (256 x 256 tiles, LDS-DMA operands, persistent workgroups): K a multiple of 64 - the token-embedding shapes
    reach it with their operands zero-padded along K (668 -> 704, 1336 -> 1344) - ragged M and N tiles, every epilogue the path uses:
    bias + saved pre-activation + softsign (tokeniser forward), GELU, dropout (same mask as mmfm_dropout_apply), residual, and the
    two backward-through-activation forms.  Against torch fp64 on the same bf16 inputs; the first rows bit for bit against the
    128 x 128 kernel (MMFM_GEMM_BIG_KMIN keeps short reductions there: here via a K = 64 * 7 = 448 < 512 twin is not possible, so
    the comparison is against fp64 only)."""
    x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    x[:, :Kreal] = bf(rnd(M, Kreal, seed=1))
    w[:, :Kreal] = bf(rnd(N, Kreal, seed=2, scale=Kreal ** -0.5))
    b = rnd(N, seed=3)
    ref0 = x.double() @ w.double().T + b.double()
    y, pre = torch.full((M + 1, N), 9.0, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    # (1) bias + pre_out + softsign
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, pre_out=pre, act=2, act_scale=0.7)
    close_bf16(pre, ref0, "256-tile pre-activation")
    close_bf16(y[:M], 0.7 * ref0 / (1 + ref0.abs()), "256-tile softsign")
    assert torch.all(y[M:] == 9.0), "rows beyond M written"
    # (2) GELU
    ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=1)
    close_bf16(y[:M], torch.nn.functional.gelu(ref0), "256-tile gelu")
    # (3) dropout + residual: the mask is the one mmfm_dropout_apply draws for (site, m * N + n)
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 5)
    res = bf(rnd(M, N, seed=4))
    y0, yd, msk = (torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N, bias=b)
    close_bf16(y0, ref0, "256-tile plain")
    ops.gemm(x, w, yd, M, N, K, lda=K, ldb=K, ldc=N, bias=b, drop=ops.dropout(state, 9, 0.4), residual=res, ldr=N)
    ops.dropout_apply(torch.ones_like(y0), msk, M, N, ops.dropout(state, 9, 0.4))
    keep = (msk != 0)
    assert abs(keep.float().mean().item() - 0.6) < 0.01
    close_bf16(yd, ref0 * keep / 0.6 + res.double(), "256-tile dropout + residual", tol=2e-2)
    # (4) backward through the activations: act 3 (gelu') and act 4 (softsign') multiply by f'(saved pre-activation)
    u = bf(rnd(M, N, seed=6))
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N, gradmul_pre=u, act=3)
    ud = u.double()
    gp = 0.5 * (1 + torch.erf(ud / math.sqrt(2))) + ud * torch.exp(-0.5 * ud * ud) / math.sqrt(2 * math.pi)
    close_bf16(y0, (x.double() @ w.double().T) * gp, "256-tile gelu'")
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N, gradmul_pre=u, act=4, act_scale=0.7)
    close_bf16(y0, (x.double() @ w.double().T) * 0.7 / (1 + ud.abs()) ** 2, "256-tile softsign'")
    # act 5: the same derivative from the activation OUTPUT a = 0.7 softsign(u) (what the bf16 engine saves instead of u)
    a_out = bf(0.7 * ud / (1 + ud.abs()))
    ops.gemm(x, w, y0, M, N, K, lda=K, ldb=K, ldc=N, gradmul_pre=a_out, act=5, act_scale=0.7)
    close_bf16(y0, (x.double() @ w.double().T) * 0.7 * (1 - a_out.double().abs() / 0.7) ** 2, "256-tile softsign' from the output")
    close_bf16(y0, (x.double() @ w.double().T) * 0.7 / (1 + ud.abs()) ** 2, "256-tile softsign' from the output vs exact", tol=2.5e-2)


def test_gemm_act5_softsign_grad_from_output_error_bound(ops):
    """act 5 rebuilds softsign'(x) = 1 / (1 + |x|)^2 as (1 - |y| / s)^2 from the bf16-STORED activation y = s x / (1 + |x|).
    Near |y| -> s this cancels: bf16 spacing there is s 2^-8, so r = 1 - |y| / s carries an ABSOLUTE error of up to 2^-9 ~ 2e-3 and the
    factor r^2 an absolute error of <= 2 r 2^-9 + 2^-18 - small against the gradients of moderate pre-activations, the whole value once
    |x| > ~250 (y rounds to s, the gradient reads exactly 0 where the true factor is 1.6e-5).  Pinned here over |x| in [0, 300]:
    absolute error of the factor <= 1.5 (2^-8 r + 2^-17) (+ the bf16 rounding of the product) everywhere, i.e. a RELATIVE error of about
    0.6 % x (1 + |x|): 5 % at |x| = 8, 20 % at |x| = 32 (checked up to there), the whole value beyond |x| ~ 170."""
    M, N, K, s_ = 64, 512, 64, 0.7
    xs = torch.cat([torch.linspace(0, 8, M * N // 2), torch.linspace(8, 300, M * N // 2)])[torch.randperm(M * N, generator=torch.Generator().manual_seed(3))]
    xs = (xs * torch.where(torch.rand(M * N, generator=torch.Generator().manual_seed(4)) < 0.5, -1.0, 1.0)).view(M, N).cuda()
    y_act = bf(s_ * xs / (1 + xs.abs()))
    a = bf(torch.eye(M, K, device="cuda"))                    # dY = a @ w^T picks columns of w: dY[m, n] = w[n, m]
    w = bf(rnd(N, K, seed=5) + 2.0 * torch.sign(rnd(N, K, seed=6)))          # |w| >= ~1: the factor is read off out / w
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, out, M, N, K, lda=K, ldb=K, ldc=N, gradmul_pre=y_act, act=5, act_scale=s_)
    base = a.double() @ w.double().T
    got = out.double() / base                                  # the applied factor, per element
    r = 1 / (1 + xs.double().abs())
    true = s_ * r * r
    err = (got - true).abs()
    bound = 1.5 * (s_ * (2.0 ** -8 * r + 2.0 ** -17) + 2.0 ** -8 * true + 2.0 ** -9 * got.abs())      # + bf16 rounding of the stored product; x1.5 slack
    assert bool((err <= bound).all()), f"max excess {(err - bound).max().item():.3e}"
    rel_bound = 1.5 * 2.0 ** -8 * (1 + xs.double().abs()) + 2.0 ** -7           # ~0.6 % x (1 + |x|): 5 % at |x| = 8, the whole value beyond ~170
    small = xs.abs() <= 32
    assert bool((err[small] / true[small] <= rel_bound[small]).all())


def _extract_attn_keep_mask(ops, state, site, p, B, heads, Lq, Lk):
    """keep[b, h, q, k] of the attention-probability dropout at (state, site), read off the kernel itself: the decisions depend on
    (state, site, b, head, query, key) only, not on the data, so with q = k = 0 (uniform probabilities 1 / Lk) and a one-hot V block
    (V[k, d] = 1 iff k == 32 blk + d) the forward output is keep(q, 32 blk + d) / (Lk (1 - p)): ceil(Lk / 32) launches show every key."""
    from multi_modal_foundation_model_amd import _lib as Lb
    dh = 32
    H = heads * dh
    q = torch.zeros(B * Lq, H, device="cuda", dtype=torch.bfloat16)
    kp = torch.ones(B, Lk, dtype=torch.uint8, device="cuda")
    keep = torch.zeros(B, heads, Lq, Lk, dtype=torch.bool, device="cuda")
    kb = torch.empty(ops.attn_keepbits_bytes(B, heads, Lq, Lk), dtype=torch.uint8, device="cuda")
    for blk in range((Lk + 31) // 32):
        kv = torch.zeros(B, Lk, 2, heads, dh, device="cuda", dtype=torch.bfloat16)
        n = min(32, Lk - 32 * blk)
        kv[:, 32 * blk + torch.arange(n), 1, :, torch.arange(n)] = 1.0
        kv = kv.view(B * Lk, 2 * H)
        o, lse = torch.empty(B * Lq, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, Lq, device="cuda")
        desc = ops.attn_desc(Lb.BF16, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                             kp, None, 0, dh ** -0.5, drop_p=ops.dropout(state, site, p), keepbits=kb)
        ops.attn_fwd(desc)
        keep[:, :, :, 32 * blk:32 * blk + n] = (o.view(B, Lq, heads, dh).permute(0, 2, 1, 3)[..., :n] != 0)
    return keep


def _unpack_keepbits(kb, B, heads, Lq, Lk):
    """keep[b, h, q, k] out of the documented bit-tile layout (csrc/attention_fast.hip header): words [bh][qt][kt][32], word 2 r + kh
    of a tile = key 32 kt + (r & 3) + 8 (r >> 2) + 4 kh, bit j = query 32 qt + j."""
    nqt, nkt = (Lq + 31) // 32, (Lk + 31) // 32
    w = kb[:B * heads * nqt * nkt * 128].view(torch.int32).view(B * heads, nqt, nkt, 32).cpu().numpy().astype(np.uint32)      # (behind the tiles: scratch)
    bits = ((w[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)          # [bh, qt, kt, word, qbit]
    widx = np.arange(32)
    key_of_word = ((widx >> 1) & 3) + 8 * (widx >> 3) + 4 * (widx & 1)
    keep = np.zeros((B * heads, nqt * 32, nkt * 32), dtype=bool)
    for qt in range(nqt):
        for kt in range(nkt):
            keep[:, 32 * qt:32 * qt + 32, 32 * kt + key_of_word] = bits[:, qt, kt].transpose(0, 2, 1)
    return torch.from_numpy(keep[:, :Lq, :Lk]).view(B, heads, Lq, Lk)


@pytest.mark.parametrize("B,heads,Lq,Lk,flags,pad", [(2, 8, 200, 200, 1, True), (2, 8, 200, 200, 0, False), (3, 4, 72, 40, 0, True), (2, 4, 224, 224, 1, False),
                                                     (2, 2, 104, 104, 1, True), (80, 8, 200, 200, 1, True)])      # 640 heads: more than a persistent grid
def test_attention_fast_dropout_matches_reference(ops, B, heads, Lq, Lk, flags, pad):
    """The dh = 32 fast pair (csrc/attention_fast.hip) WITH attention-probability dropout against torch fp32 on the same bf16 inputs:
    the keep mask is read off the kernel (see _extract_attn_keep_mask), then softmax -> mask / (1 - p) -> P.V and its autograd give
    the expected output, LSE and all three gradients (forward and backward must regenerate the SAME decisions).  Keys of the last
    quarter carry 6x larger rows so that the running maximum jumps late in the key sweep by more than the lazy-rescale threshold
    (cdna_hip_programming.md rule 26: force the rare branch)."""
    from multi_modal_foundation_model_amd import _lib as Lb
    dh, p = 32, 0.4
    H = heads * dh
    q = bf(rnd(B * Lq, H, seed=1))
    kv = rnd(B * Lk, 2 * H, seed=2)
    kv.view(B, Lk, 2 * H)[:, (3 * Lk) // 4:, :H] *= 6.0
    kv = bf(kv)
    d_o = bf(rnd(B * Lq, H, seed=3))
    kp = torch.ones(B, Lk, dtype=torch.uint8)
    if pad:
        kp[0, Lk - 3:] = 0
        kp[B - 1, 5:9] = 0
    kp = kp.cuda()
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 4321)
    keep = _extract_attn_keep_mask(ops, state, 7, p, B, heads, Lq, Lk)
    keep_p = ops.attn_keep_prob(p)                       # the keep-bit path honours p to 2^-10 and scales survivors by 1 / keep_p
    assert abs(keep_p - (1 - p)) <= 2 ** -11
    rate = keep.float().mean().item()
    assert abs(rate - keep_p) < 5 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-4, f"keep rate {rate}"
    o, lse = torch.empty(B * Lq, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, Lq, device="cuda")
    dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
    kb = torch.empty(ops.attn_keepbits_bytes(B, heads, Lq, Lk), dtype=torch.uint8, device="cuda")
    desc = ops.attn_desc(Lb.BF16, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, flags, dh ** -0.5, drop_p=ops.dropout(state, 7, p), d_o=d_o.data_ptr(), lddo=H, dq=dq.data_ptr(),
                         dk=dkv.data_ptr(), dv=dkv.data_ptr() + H * 2, lddq=H, lddk=2 * H, lddv=2 * H, keepbits=kb)
    ops.attn_fwd(desc)
    # the bit tiles the forward left for the backward are the decisions the forward itself applied (documented layout)
    assert torch.equal(_unpack_keepbits(kb, B, heads, Lq, Lk), keep.cpu())
    ops.attn_bwd(desc)
    m = kp.bool()[:, None, :].expand(B, Lq, Lk)
    if flags & 1:
        m = m | torch.eye(Lq, dtype=torch.bool, device="cuda")[None]
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    Q = qr.view(B, Lq, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, Lk, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    s = (Q @ K_.transpose(-1, -2)) * dh ** -0.5
    s = s.masked_fill(~m[:, None], float("-inf"))
    P = torch.softmax(s, -1)
    oref = ((P * keep.float() / keep_p) @ V_).transpose(1, 2).reshape(B * Lq, H)
    close_bf16(o, oref, "fast attn fwd with dropout", tol=2e-2)
    close(lse, torch.logsumexp(s, -1), rtol=1e-3, atol=2e-3, msg="fast attn lse")
    oref.backward(d_o.float())
    close_bf16(dq, qr.grad, "fast attn dq", tol=3e-2)
    close_bf16(dkv[:, :H], kvr.grad[:, :H], "fast attn dk", tol=3e-2)
    close_bf16(dkv[:, H:], kvr.grad[:, H:], "fast attn dv", tol=3e-2)


@pytest.mark.parametrize("B,heads,Lq,Lk,flags,pad,p", [(2, 2, 600, 600, 1, True, 0.4), (1, 2, 200, 200, 0, False, 0.4), (2, 1, 72, 136, 0, True, 0.0),
                                                      (1, 2, 608, 600, 0, False, 0.25), (1, 1, 40, 24, 0, True, 0.4)])
def test_attention_long_keepbit_kernels_match_reference(ops, B, heads, Lq, Lk, flags, pad, p):
    """The dh = 64 keep-bit kernels (csrc/attention_long.hip: BASELINE configs[4], L = 600) against torch fp32 on the same bf16 inputs:
    chunk-streamed forward with the first-tile reference exponent, prep + dK/dV + dQ backward.  The keep mask is read out of the bit
    tiles the forward left (layout pinned by test_attention_fast_dropout_matches_reference against the dh = 32 forward's own output),
    then softmax -> mask / keep -> P.V and its autograd give the expected output, LSE and all three gradients.  Keys of the last
    quarter carry 6x larger rows (scores far above the first key tile's maximum)."""
    from multi_modal_foundation_model_amd import _lib as Lb
    dh = 64
    H = heads * dh
    q = bf(rnd(B * Lq, H, seed=1))
    kv = rnd(B * Lk, 2 * H, seed=2)
    kv.view(B, Lk, 2 * H)[:, (3 * Lk) // 4:, :H] *= 6.0
    kv = bf(kv)
    d_o = bf(rnd(B * Lq, H, seed=3))
    kp = torch.ones(B, Lk, dtype=torch.uint8)
    if pad:
        kp[0, Lk - 3:] = 0
        kp[B - 1, 5:9] = 0
    kp = kp.cuda()
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 99)
    keep_p = ops.attn_keep_prob(p)
    o, lse = torch.empty(B * Lq, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, Lq, device="cuda")
    dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
    kb = torch.zeros(ops.attn_keepbits_bytes(B, heads, Lq, Lk), dtype=torch.uint8, device="cuda")
    desc = ops.attn_desc(Lb.BF16, B, heads, Lq, Lk, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, flags, dh ** -0.5, drop_p=ops.dropout(state, 5, p), drop_o=ops.dropout(state, 6, 0.0), d_o=d_o.data_ptr(), lddo=H,
                         dq=dq.data_ptr(), dk=dkv.data_ptr(), dv=dkv.data_ptr() + H * 2, lddq=H, lddk=2 * H, lddv=2 * H, keepbits=kb)
    ops.attn_fwd(desc)
    keep = _unpack_keepbits(kb, B, heads, Lq, Lk).cuda() if p > 0 else torch.ones(B, heads, Lq, Lk, dtype=torch.bool, device="cuda")
    if p > 0:
        rate = keep.float().mean().item()
        assert abs(rate - keep_p) < 5 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-4, f"keep rate {rate}"
    ops.attn_bwd(desc)
    m = kp.bool()[:, None, :].expand(B, Lq, Lk)
    if flags & 1:
        m = m | torch.eye(Lq, Lk, dtype=torch.bool, device="cuda")[None]
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    Q = qr.view(B, Lq, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, Lk, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    s = (Q @ K_.transpose(-1, -2)) * dh ** -0.5
    s = s.masked_fill(~m[:, None], float("-inf"))
    P = torch.softmax(s, -1)
    oref = ((P * keep.float() / keep_p) @ V_).transpose(1, 2).reshape(B * Lq, H)
    close_bf16(o, oref, "long attn fwd", tol=2e-2)
    close(lse, torch.logsumexp(s, -1), rtol=1e-3, atol=2e-3, msg="long attn lse")
    oref.backward(d_o.float())
    close_bf16(dq, qr.grad, "long attn dq", tol=3e-2)
    close_bf16(dkv[:, :H], kvr.grad[:, :H], "long attn dk", tol=3e-2)
    close_bf16(dkv[:, H:], kvr.grad[:, H:], "long attn dv", tol=3e-2)


def test_attention_long_exact_pass_on_large_scores(ops):
    """dh = 64 forward (csrc/attention_long.hip): rows whose sum overflows against the first key tile's maximum send the WORKGROUP (a vote
    through LDS: the chunk barriers need every wave) through the exact pass.  Keys 32.. carry 40x larger rows, one head's first tile is
    fully padded; Lq spans two workgroups (8 query tiles each) so that a redo and a no-redo workgroup run side by side."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads, L, dh = 2, 2, 328, 64
    H = heads * dh
    q = bf(rnd(B * L, H, seed=21) * 2.0)
    kv = rnd(B * L, 2 * H, seed=22)
    kv.view(B, L, 2 * H)[:, 32:, :H] *= 40.0
    q.view(B, L, H)[:, 256:] *= 0.0                                      # the second workgroup's rows: flat scores, no overflow, no redo
    kv = bf(kv)
    d_o = bf(rnd(B * L, H, seed=23))
    kp = torch.ones(B, L, dtype=torch.uint8)
    kp[1, :32] = 0
    kp = kp.cuda()
    o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
    dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
    kb = torch.zeros(ops.attn_keepbits_bytes(B, heads, L, L), dtype=torch.uint8, device="cuda")
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, 0, dh ** -0.5, d_o=d_o.data_ptr(), lddo=H, dq=dq.data_ptr(), dk=dkv.data_ptr(), dv=dkv.data_ptr() + H * 2,
                         lddq=H, lddk=2 * H, lddv=2 * H, keepbits=kb)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    Q = qr.view(B, L, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, L, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    s = (Q @ K_.transpose(-1, -2)) * dh ** -0.5
    assert s.max().item() > 150
    s = s.masked_fill(~kp.bool()[:, None, None, :], float("-inf"))
    oref = (torch.softmax(s, -1) @ V_).transpose(1, 2).reshape(B * L, H)
    close_bf16(o, oref, "long exact-pass fwd", tol=2e-2)
    close(lse, torch.logsumexp(s, -1), rtol=1e-3, atol=2e-2, msg="long exact-pass lse")
    oref.backward(d_o.float())
    close_bf16(dq, qr.grad, "long exact-pass dq", tol=3e-2)
    close_bf16(dkv[:, :H], kvr.grad[:, :H], "long exact-pass dk", tol=3e-2)
    close_bf16(dkv[:, H:], kvr.grad[:, H:], "long exact-pass dv", tol=3e-2)


def test_attention_fast_exact_pass_on_large_scores(ops):
    """The fast forward keeps the first key tile's row maximum as the reference exponent for the whole row (no running maximum); a
    row whose later scores overflow against it must come out of the exact pass instead.  Keys 32.. carry 40x larger rows than the
    first 32 (scores ~ +-500 against ~ +-12), one head's first tile is fully padded: LSE, output and gradients against torch fp32."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads, L, dh = 2, 4, 200, 32
    H = heads * dh
    q = bf(rnd(B * L, H, seed=11) * 2.0)
    kv = rnd(B * L, 2 * H, seed=12)
    kv.view(B, L, 2 * H)[:, 32:, :H] *= 40.0
    kv = bf(kv)
    d_o = bf(rnd(B * L, H, seed=13))
    kp = torch.ones(B, L, dtype=torch.uint8)
    kp[1, :32] = 0
    kp = kp.cuda()
    o, lse = torch.empty(B * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(B, heads, L, device="cuda")
    dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
    desc = ops.attn_desc(Lb.BF16, B, heads, L, L, dh, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, H, 2 * H, 2 * H, o.data_ptr(), H, lse,
                         kp, None, 0, dh ** -0.5, d_o=d_o.data_ptr(), lddo=H, dq=dq.data_ptr(), dk=dkv.data_ptr(), dv=dkv.data_ptr() + H * 2,
                         lddq=H, lddk=2 * H, lddv=2 * H)
    ops.attn_fwd(desc)
    ops.attn_bwd(desc)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    Q = qr.view(B, L, heads, dh).transpose(1, 2)
    K_, V_ = [t.view(B, L, heads, dh).transpose(1, 2) for t in kvr.split(H, dim=1)]
    s = (Q @ K_.transpose(-1, -2)) * dh ** -0.5
    assert s.max().item() > 150                                        # beyond 2^100 against a first-tile reference
    s = s.masked_fill(~kp.bool()[:, None, None, :], float("-inf"))
    oref = (torch.softmax(s, -1) @ V_).transpose(1, 2).reshape(B * L, H)
    close_bf16(o, oref, "exact-pass fwd", tol=2e-2)
    close(lse, torch.logsumexp(s, -1), rtol=1e-3, atol=2e-2, msg="exact-pass lse")
    oref.backward(d_o.float())
    close_bf16(dq, qr.grad, "exact-pass dq", tol=3e-2)
    close_bf16(dkv[:, :H], kvr.grad[:, :H], "exact-pass dk", tol=3e-2)
    close_bf16(dkv[:, H:], kvr.grad[:, H:], "exact-pass dv", tol=3e-2)


def test_attention_fast_dropout_statistics(ops):
    """Keep decisions of the fast attention pair at the bench head shape: rate, and no correlation between neighbouring queries / keys,
    the two decisions of one hash (keys 2j, 2j + 1), heads, samples, sites and steps (|corr| < 0.01; noise ~1e-3 on 1.3 M bits)."""
    B, heads, L, p = 4, 8, 200, 0.4
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 77)
    k = _extract_attn_keep_mask(ops, state, 3, p, B, heads, L, L).float()
    var = p * (1 - p)
    keep_p = ops.attn_keep_prob(p)
    assert abs(k.mean().item() - keep_p) < 4 * math.sqrt(var / k.numel())
    c = k - keep_p
    corr = lambda a, b: (a * b).mean().item() / var
    assert abs(corr(c[..., 0::2], c[..., 1::2])) < 0.01                                   # neighbouring keys
    for lag in (1, 2, 16, 32):
        assert abs(corr(c[:, :, :-lag], c[:, :, lag:])) < 0.01, f"query lag {lag}"
        assert abs(corr(c[:, :, :, :-lag], c[:, :, :, lag:])) < 0.01, f"key lag {lag}"
    assert abs(corr(c[:, :-1], c[:, 1:])) < 0.01 and abs(corr(c[:-1], c[1:])) < 0.01      # heads, samples
    assert abs(corr(c, _extract_attn_keep_mask(ops, state, 4, p, B, heads, L, L).float() - keep_p)) < 0.01          # another site
    assert torch.equal(k, _extract_attn_keep_mask(ops, state, 3, p, B, heads, L, L).float())
    ops.rng_advance(state)
    assert abs(corr(c, _extract_attn_keep_mask(ops, state, 3, p, B, heads, L, L).float() - keep_p)) < 0.01          # next step


@pytest.mark.parametrize("M,N,K,kc", [(204800 + 37, 768, 256, 1), (204800 + 37, 256, 768, 0), (204800, 512, 256, 1)])
def test_gemm_bf16_full_size_many_tiles_per_workgroup(ops, M, N, K, kc):
    """BASELINE configs[1] row count (B = 1024 x 200 tokens): 4,800-9,600 output tiles, so the persistent bf16-output kernel walks
    several tiles per workgroup with the next tile's first K-slice prefetched during the epilogue.  Sampled rows against torch fp64,
    and the first rows bit for bit against a launch small enough to be one tile per workgroup."""
    x, b = bf(rnd(M, K, seed=1)), rnd(N, seed=3)
    w = bf(rnd(N, K, seed=2, scale=K ** -0.5)) if kc else bf(rnd(K, N, seed=2, scale=K ** -0.5))
    y = torch.full((M + 2, N), 9.0, device="cuda", dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    idx = torch.cat([torch.arange(0, 300), torch.randint(0, M, (2048,), generator=g), torch.arange(M - 300, M)]).unique().cuda()
    if kc:
        ops.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b)
        ref = x[idx].double() @ w.double().T + b.double()
    else:
        ops.gemm(x, w, y, M, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0)
        ref = x[idx].double() @ w.double()
    close_bf16(y[idx], ref, f"bf16 full-size {M}x{N}x{K}")
    assert torch.all(y[M:] == 9.0)
    ys = torch.empty(1024, N, device="cuda", dtype=torch.bfloat16)
    if kc:
        ops.gemm(x[:1024], w, ys, 1024, N, K, lda=K, ldb=K, ldc=N, bias=b)
    else:
        ops.gemm(x[:1024], w, ys, 1024, N, K, lda=K, ldb=N, ldc=N, b_kcontig=0)
    assert torch.equal(ys, y[:1024])


def test_attention_bf16_full_batch_matches_small_launches(ops):
    """BASELINE configs[1] batch (B = 1024, 8 heads, L = 200, dh = 32): 8,192 workgroups in the XCD-aware (batch, head) order.  Without
    dropout a head's result does not depend on where it sits in the batch: samples from the start, middle and end of the batch must
    equal, bit for bit, a launch over those samples alone (forward output, LSE and all three gradients), and sample 0 is also
    checked against the fp32 reference."""
    from multi_modal_foundation_model_amd import _lib as Lb
    B, heads, L, dh = 1024, 8, 200, 32
    H = heads * dh
    qkv, d_o = bf(rnd(B * L, 3 * H, seed=1)), bf(rnd(B * L, H, seed=2))
    keypad = torch.ones(B, L, dtype=torch.uint8)
    keypad[0, L - 3:] = 0
    keypad[B - 1, 5:9] = 0
    kp, mi = keypad.cuda(), (torch.arange(L) >= L // 2).to(torch.uint8).cuda()
    scale, es = 1.0 / math.sqrt(dh), 2

    def run(qkv_, d_o_, kp_, nb):
        o, lse = torch.empty(nb * L, H, device="cuda", dtype=torch.bfloat16), torch.empty(nb, heads, L, device="cuda")
        dqkv = torch.full((nb * L, 3 * H), float("nan"), device="cuda", dtype=torch.bfloat16)
        base = qkv_.data_ptr()
        desc = ops.attn_desc(Lb.BF16, nb, heads, L, L, dh, base, base + H * es, base + 2 * H * es, 3 * H, 3 * H, 3 * H, o.data_ptr(), H, lse, kp_, mi,
                             1, scale, d_o=d_o_.data_ptr(), lddo=H, dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + H * es,
                             dv=dqkv.data_ptr() + 2 * H * es, lddq=3 * H, lddk=3 * H, lddv=3 * H)
        ops.attn_fwd(desc)
        ops.attn_bwd(desc)
        torch.cuda.synchronize()
        return o, lse, dqkv

    o, lse, dqkv = run(qkv, d_o, kp, B)
    for b0 in (0, 511, B - 2):
        rows = slice(b0 * L, (b0 + 2) * L)
        o2, lse2, dq2 = run(qkv[rows].contiguous(), d_o[rows].contiguous(), kp[b0:b0 + 2].contiguous(), 2)
        assert torch.equal(o[rows], o2) and torch.equal(lse[b0:b0 + 2], lse2) and torch.equal(dqkv[rows], dq2), f"samples {b0}, {b0 + 1}"
    x = qkv[:L].float().requires_grad_(True)
    q, k, v = [t.view(1, L, heads, dh).transpose(1, 2) for t in x.split(H, dim=1)]
    m = kp[:1].bool()[:, None, :].expand(1, L, L) | torch.eye(L, dtype=torch.bool, device="cuda")[None]
    oref = ref_attention(q, k, v, m, scale).transpose(1, 2).reshape(L, H)
    close_bf16(o[:L], oref, "bf16 attn fwd, sample 0 of 1024", tol=2e-2)
    oref.backward(d_o[:L].float())
    close_bf16(dqkv[:L], x.grad, "bf16 attn grads, sample 0 of 1024", tol=3e-2)


def test_dropout_pair_hash_statistics(ops):
    """The element dropout mask (one hash per PAIR of neighbouring elements, 16 bits each): keep rate within 4 sigma of 1 - p, no
    correlation between the two elements of a pair, between neighbouring pairs, between rows, between sites or between steps
    (|corr| < 0.01 on 4 M elements, noise 5e-4)."""
    M, N, p = 4096, 1024, 0.4
    x = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 1234)

    def mask(site):
        y = torch.empty_like(x)
        ops.dropout_apply(x, y, M, N, ops.dropout(state, site, p))
        return (y != 0).float()

    k = mask(3)
    assert abs(k.mean().item() - (1 - p)) < 4 * math.sqrt(p * (1 - p) / (M * N))
    c = k - (1 - p)
    var = p * (1 - p)
    corr = lambda a, b: (a * b).mean().item() / var
    assert abs(corr(c[:, 0::2], c[:, 1::2])) < 0.01                      # the two halves of one hash
    for lag in (1, 2, 3, 4, 8, 16, 32, 64):
        assert abs(corr(c[:, :-lag], c[:, lag:])) < 0.01, f"column lag {lag}"
    for lag in (1, 2, 8):
        assert abs(corr(c[:-lag], c[lag:])) < 0.01, f"row lag {lag}"
    assert abs(corr(c, mask(4) - (1 - p))) < 0.01                        # another site
    assert torch.equal(k, mask(3))                                       # same state, same site: same mask
    ops.rng_advance(state)
    assert abs(corr(c, mask(3) - (1 - p))) < 0.01                        # next step


def test_dropout_hash_large_tensor_has_no_repeated_masks(ops):
    """the 24-bit multiply of the pair hash dropped input bits 24..31, so element idx and idx ^ (1 << 25 | 1 << 9)
    shared a decision and a [204800, 256] tensor repeated its mask at row lag 131072 (rows r and (r - 131072) ^ 2).  On a 2^26-element
    tensor: no correlation at row lags 131070..131074 and no exact equality between the two halves related by that XOR."""
    M, N, p = 262144, 256, 0.4                      # 2^26 elements
    x = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
    state = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.rng_seed(state, 99)
    y = torch.empty_like(x)
    ops.dropout_apply(x, y, M, N, ops.dropout(state, 5, p))
    k = (y != 0)
    del x, y
    assert abs(k.float().mean().item() - (1 - p)) < 4 * math.sqrt(p * (1 - p) / (M * N))
    var = p * (1 - p)
    c = k.float() - (1 - p)
    for lag in (131070, 131071, 131072, 131073, 131074):
        assert abs((c[:-lag] * c[lag:]).mean().item() / var) < 0.01, f"row lag {lag}"
    # idx ^ (1 << 25 | 1 << 9): row ^ 131072, column pair ^ 2 (row index bit 17, element bit 9 = row bit 1 at N = 256)
    flat = k.view(-1)
    idx = torch.arange(0, 1 << 22, device="cuda", dtype=torch.int64) * 16 + 3
    agree = (flat[idx] == flat[idx ^ ((1 << 25) | (1 << 9))]).float().mean().item()
    assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01, agree           # independent masks agree 52 % of the time, copies 100 %


def test_gemm_bf16_gelu_polynomial_accuracy(ops):
    """The bf16 kernels' GELU (packed degree-9 polynomial Phi, common.h phi2) and its derivative through the GEMM epilogues with an
    identity weight: |gelu error| <= 1 bf16 ulp of the result + 5e-5 on [-4, 4], and |x| * 4e-5 beyond (Phi saturates at 1 - 3e-5)."""
    n = 256
    xs = torch.linspace(-9.0, 9.0, 64 * n, device="cuda").view(64, n)
    x = bf(xs)
    eye = torch.eye(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(64, n, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, eye, y, 64, n, n, lda=n, ldb=n, ldc=n, act=1)
    xd = x.double()
    ref = torch.nn.functional.gelu(xd)
    err = (y.double() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 5e-5 + xd.abs() * 4e-5
    assert torch.all(err <= bound), f"gelu: worst excess {(err - bound).max().item():.3e} at x = {xd.flatten()[(err - bound).argmax()].item():.4f}"
    # derivative: dx = dy * gelu'(pre) with dy = 1 (K = n identity product of ones-rows is the row sum: use dy = e_j rows instead)
    dy = torch.eye(n, device="cuda", dtype=torch.bfloat16)[:64].contiguous()          # row i = e_i  ->  (dy @ I)[i, j] = delta_ij
    pre = bf(torch.linspace(-9.0, 9.0, 64, device="cuda")[:, None].expand(64, n).contiguous())
    dx = torch.empty(64, n, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, eye, dx, 64, n, n, lda=n, ldb=n, ldc=n, b_kcontig=0, act=3, gradmul_pre=pre)
    p64 = pre.double()[:, 0].clone().requires_grad_(True)
    torch.nn.functional.gelu(p64).sum().backward()
    got = dx.double()[torch.arange(64), torch.arange(64)]
    assert torch.all((got - p64.grad).abs() <= p64.grad.abs() * 2.0 ** -8 + 1e-4), f"gelu': {(got - p64.grad).abs().max().item():.3e}"


def test_reduce_slabs_multi_matches_single_reductions(ops):
    """mmfm_reduce_slabs_multi: several slab reductions (ragged sizes, different slab counts, one accumulating) in one launch."""
    specs = [(1000, 3, False), (256, 1, False), (77, 15, True), (4096 + 5, 7, False), (1, 2, False)]
    items, refs = [], []
    for i, (n, S, acc) in enumerate(specs):
        stride = (n + 7) // 8 * 8
        src = rnd(S, stride, seed=10 + i)
        dst = rnd(n, seed=50 + i)
        refs.append((dst.double() if acc else 0) + src[:, :n].double().sum(0))
        items.append((dst, src, n, S, stride, acc))
    ops.reduce_slabs_multi(items, "cuda")
    for (dst, *_), ref in zip(items, refs):
        close(dst, ref, rtol=1e-5, atol=1e-5, msg="reduce_slabs_multi")


def test_gemm_pair_two_weight_gradients_in_one_launch(ops):
    """mmfm_gemm_pair: the MLP's two weight gradients (512x256 and 256x512 over R rows, column sums riding along) from ONE launch of the
    streaming kernel, each on its share of the CUs, against the same two products issued separately and against fp64."""
    R = 51200
    dy1, x1 = bf(rnd(R, 512, seed=31)), bf(rnd(R, 256, seed=32))
    dy2, x2 = bf(rnd(R, 256, seed=33)), bf(rnd(R, 512, seed=34))

    def run(pair):
        outs = []
        descs = []
        for dy, x, S in ((dy1, x1, 32), (dy2, x2, 32)):
            N, K = dy.shape[1], x.shape[1]
            kchunk = (-(-R // S) + 63) // 64 * 64
            S2 = -(-R // kchunk)
            stride = (N * K + N + 7) // 8 * 8
            slabs = torch.full((S2, stride), float("nan"), device="cuda")
            kw = dict(lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0, splits=S2, kchunk=kchunk, slab_stride=stride, c_f32=1, colsum=slabs.data_ptr() + 4 * N * K)
            descs.append((ops.gemm_desc(dy, x, slabs, N, K, R, **kw), slabs, N, K, S2, stride))
        if pair:
            ops.gemm_pair(descs[0][0], descs[1][0])
        else:
            from multi_modal_foundation_model_amd import _lib as L
            import ctypes as C
            for d in descs:
                assert L.lib().mmfm_gemm(C.byref(d[0]), None) == 0
        for _, slabs, N, K, S2, stride in descs:
            out = torch.empty(N * K + N, device="cuda")
            ops.reduce_slabs(out, slabs, N * K + N, S2, stride)
            outs.append(out)
        return outs
    a, b = run(True), run(False)
    for o, r, (dy, x) in zip(a, b, ((dy1, x1), (dy2, x2))):
        assert torch.equal(o, r)                    # same kernel, same split: bit-identical to the separate launches
        N, K = dy.shape[1], x.shape[1]
        close_bf16(o[:N * K].view(N, K), dy.double().T @ x.double(), "paired dW", tol=2e-3)
        close(o[N * K:], dy.double().sum(0).float(), rtol=1e-5, atol=1e-4 * math.sqrt(R), msg="paired bias grad")

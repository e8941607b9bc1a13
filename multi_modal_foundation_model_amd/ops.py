"""Tensor-level wrappers over the C-ABI.  Each op either runs now (on torch's current stream) or,
when `plan` is a list, is appended to it as a pre-bound `(cfunc, args)` pair: the engine builds such
a plan once per batch shape and replays it every step (no per-step Python marshalling, and the
replay is hipGraph-capturable because no call allocates or synchronises)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _stream():
    return torch.cuda.current_stream().cuda_stream


def run_plan(plan, stream=None):
    st = _stream() if stream is None else stream
    for fn, args, _keep in plan:
        rc = fn(*args, st)
        if rc:
            L.check(rc, fn.__name__)


def _emit(plan, fn, args, keep=()):
    if plan is None:
        L.check(fn(*args, _stream()), fn.__name__)
    else:
        plan.append((fn, args, keep))


def P(t):
    return None if t is None else t.data_ptr()


def dt(t):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def dropout(state, site, p):
    if state is None or p <= 0:
        return L.NO_DROP
    return L.Dropout(state.data_ptr(), site, float(p))


def gemm_desc(A, B, Cout, M, N, K, *, lda, ldb, ldc, a_kcontig=1, b_kcontig=1, bias=None, pre_out=None, act=0,
              act_scale=1.0, gradmul_pre=None, drop=None, residual=None, ldr=0, splits=1, kchunk=0, slab_stride=0,
              dtype=None, c_f32=0, colsum=None):
    d = L.GemmDesc()
    d.dtype = dt(A) if dtype is None else dtype
    d.c_f32 = c_f32
    d.A, d.B, d.C = P(A), P(B), P(Cout)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.a_kcontig, d.b_kcontig = a_kcontig, b_kcontig
    d.splits, d.kchunk, d.slab_stride = splits, kchunk, slab_stride
    d.bias, d.pre_out, d.act, d.act_scale, d.gradmul_pre = P(bias), P(pre_out), act, act_scale, P(gradmul_pre)
    d.drop = drop if drop is not None else L.NO_DROP
    d.residual, d.ldr = P(residual), ldr
    d.colsum = colsum if isinstance(colsum, int) else P(colsum)
    return d


def gemm(A, B, Cout, M, N, K, *, plan=None, **kw):
    d = gemm_desc(A, B, Cout, M, N, K, **kw)
    _emit(plan, L.lib().mmfm_gemm, (C.byref(d),), keep=(d,))


def gemm_pair(da, db, plan=None):
    """Two independent products (descriptors from gemm_desc) through mmfm_gemm_pair: two streaming weight-gradient launches become one."""
    _emit(plan, L.lib().mmfm_gemm_pair, (C.byref(da), C.byref(db)), keep=(da, db))


def reduce_slabs(dst, src, n, nslabs, stride, accumulate=False, plan=None):
    _emit(plan, L.lib().mmfm_reduce_slabs, (P(dst), P(src), n, nslabs, stride, int(accumulate)))


def reduce_slabs_multi(items, device, plan=None):
    """items: (dst tensor, slab tensor, n, nslabs, stride, accumulate) -> one launch (mmfm_reduce_slabs_multi).  The table tensor
    is kept alive by the plan entry."""
    import numpy as np
    arr = (L.ReduceEntry * len(items))()
    chunk0 = 0
    for a, (dst, src, n, nslabs, stride, acc) in zip(arr, items):
        a.dst, a.src, a.n, a.slab_stride, a.nslabs, a.accumulate, a.chunk0 = P(dst), P(src), n, stride, nslabs, int(acc), chunk0
        chunk0 += (n + 255) // 256
    table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
    _emit(plan, L.lib().mmfm_reduce_slabs_multi, (P(table), len(items), chunk0), keep=(table,))


def colsum(x, R, N, ld, out, ws, accumulate=False, plan=None):
    _emit(plan, L.lib().mmfm_colsum, (dt(x), P(x), R, N, ld, P(out), int(accumulate), P(ws), ws.numel() * ws.element_size()))


def layernorm_fwd(x, gamma, beta, y, mean, rstd, R, H, eps=1e-5, ds_L=0, ds_T=0, plan=None):
    _emit(plan, L.lib().mmfm_layernorm_fwd, (dt(x), P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), R, H, eps, ds_L, ds_T))


def layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, R, H, ws, accumulate=False, ds_L=0, ds_T=0, plan=None):
    _emit(plan, L.lib().mmfm_layernorm_bwd, (dt(x), P(dy), P(x), P(mean), P(rstd), P(gamma), P(dres), P(dx), P(dgamma),
                                             P(dbeta), int(accumulate), R, H, ds_L, ds_T, P(ws), ws.numel() * ws.element_size()))


def attn_desc(dtype, B, heads, Lq, Lk, dh, q, k, v, ldq, ldk, ldv, o, ldo, lse, keypad, mod_id, flags, scale,
              drop_p=None, drop_o=None, d_o=None, lddo=0, dq=None, dk=None, dv=None, lddq=0, lddk=0, lddv=0, keepbits=None):
    """keepbits: uint8 tensor of attn_keepbits_bytes(B, heads, Lq, Lk) bytes (one per attention site: the forward writes the keep
    decisions of drop_p there, the backward reads them) or None (both directions hash; include/mmfm.h)."""
    d = L.AttnDesc()
    d.dtype, d.B, d.heads, d.Lq, d.Lk, d.dh = dtype, B, heads, Lq, Lk, dh
    d.q, d.k, d.v, d.ldq, d.ldk, d.ldv = q, k, v, ldq, ldk, ldv
    d.o, d.ldo, d.lse, d.keypad, d.mod_id, d.flags, d.scale = o, ldo, P(lse), P(keypad), P(mod_id), flags, scale
    d.drop_p = drop_p if drop_p is not None else L.NO_DROP
    d.drop_o = drop_o if drop_o is not None else L.NO_DROP
    d.d_o, d.lddo, d.dq, d.dk, d.dv, d.lddq, d.lddk, d.lddv = d_o, lddo, dq, dk, dv, lddq, lddk, lddv
    d.keepbits = P(keepbits)
    return d


def attn_keepbits_bytes(B, heads, Lq, Lk):
    return int(L.lib().mmfm_attn_keepbits_bytes(B, heads, Lq, Lk))


def attn_keep_prob(p):
    """Keep probability the keep-bit attention path applies for drop probability p (quantised to 2^-10)."""
    return float(L.lib().mmfm_attn_keep_prob(float(p)))


def attn_fwd(desc, plan=None):
    _emit(plan, L.lib().mmfm_attn_fwd, (C.byref(desc),), keep=(desc,))


def attn_bwd(desc, plan=None):
    _emit(plan, L.lib().mmfm_attn_bwd, (C.byref(desc),), keep=(desc,))


def mask_prep(B, T, masks, strides, attn, channels, tokmask, keypad, keep0, mod_id, count, plan=None):
    M = len(masks)
    src = (C.c_void_p * M)(*[m.data_ptr() for m in masks])
    st = (C.c_int64 * M)(*strides)
    ch = (C.c_int64 * M)(*channels)
    _emit(plan, L.lib().mmfm_mask_prep, (B, T, M, src, st, P(attn), ch, P(tokmask), P(keypad), P(keep0), P(mod_id), P(count)),
          keep=(src, st, ch))


def stitch_fwd(tok, mod_row, pos, ts, keep0, x, emb, B, T, Lseq, m, H, max_F, plan=None):
    _emit(plan, L.lib().mmfm_stitch_fwd, (dt(tok), P(tok), P(mod_row), P(pos), P(ts), P(keep0), P(x), P(emb), B, T, Lseq, m, H, max_F))


def stitch_bwd(dx, dextra, ts, keep0, drop, d_tok, d_mod_row, d_pos, acc_mod, acc_pos, B, T, Lseq, m, H, max_F, ws, plan=None):
    _emit(plan, L.lib().mmfm_stitch_bwd, (dt(dx), P(dx), P(dextra), P(ts), P(keep0), drop if drop is not None else L.NO_DROP,
                                          P(d_tok), P(d_mod_row), P(d_pos), int(acc_mod), int(acc_pos), B, T, Lseq, m, H, max_F, P(ws),
                                          ws.numel() * ws.element_size()))


def masked_loss_fwd(kind, pred, target, rowmask, mask_ld, T, R, N, loss_sum, ws, plan=None):
    _emit(plan, L.lib().mmfm_masked_loss_fwd, (dt(pred), kind, P(pred), P(target), P(rowmask), mask_ld, T, R, N, P(loss_sum), P(ws),
                                               ws.numel() * ws.element_size()))


def loss_finalize(loss_sum, count, M, loss, inv_n, plan=None):
    _emit(plan, L.lib().mmfm_loss_finalize, (P(loss_sum), P(count), M, P(loss), P(inv_n)))


def masked_loss_bwd(kind, pred, target, rowmask, mask_ld, T, R, N, grad_out, inv_n, dpred, plan=None):
    _emit(plan, L.lib().mmfm_masked_loss_bwd, (dt(pred), kind, P(pred), P(target), P(rowmask), mask_ld, T, R, N, P(grad_out), P(inv_n), P(dpred)))


def dropout_apply(src, dst, R, N, drop, plan=None):
    _emit(plan, L.lib().mmfm_dropout_apply, (dt(src), P(src), P(dst), R, N, drop))


def cast_bf16(src, dst, n, plan=None):
    _emit(plan, L.lib().mmfm_cast_f32_to_bf16, (P(src), P(dst), n))


def adamw_step(p, g, m, v, p_bf16, n, hyper, plan=None):
    _emit(plan, L.lib().mmfm_adamw_step, (P(p), P(g), P(m), P(v), P(p_bf16), n, P(hyper)))


def rng_seed(state, seed, plan=None):
    _emit(plan, L.lib().mmfm_rng_seed, (P(state), int(seed) & (2 ** 64 - 1)))


def rng_advance(state, plan=None):
    _emit(plan, L.lib().mmfm_rng_advance, (P(state),))


def collate_csr(B, max_T, max_N, pad_value, data, indices, indptr, indptr_off, nnz_off, T_b, N_b, out, tmask, smask, plan=None):
    _emit(plan, L.lib().mmfm_collate_csr, (B, max_T, max_N, float(pad_value), P(data), P(indices), P(indptr), P(indptr_off), P(nnz_off),
                                           P(T_b), P(N_b), P(out), P(tmask), P(smask)))


# ---------------------------------------------------------------------------------------------- row-owner fused kernels
def prep_table(entries, device):
    """entries: dicts with W (fp32 [N,K]) and optional gamma, beta, bias, Wp, WpT, bp, WpP, WpTP tensors -> (device table, n, tiles, keep)."""
    import numpy as np
    arr = (L.PrepEntry * len(entries))()
    tile0 = 0
    for i, e in enumerate(entries):
        N, Kd = e["W"].shape
        if e.get("WpP") is not None and Kd % 16:
            raise ValueError(f"prep_table: entry {i}: WpP (unit-permuted along K) needs K % 16 == 0, got K = {Kd} (include/mmfm.h)")
        if e.get("WpTP") is not None and N % 16:
            raise ValueError(f"prep_table: entry {i}: WpTP (unit-permuted along N) needs N % 16 == 0, got N = {N} (include/mmfm.h)")
        a = arr[i]
        a.W, a.gamma, a.beta, a.bias = P(e["W"]), P(e.get("gamma")), P(e.get("beta")), P(e.get("bias"))
        a.Wp, a.WpT, a.bp = P(e.get("Wp")), P(e.get("WpT")), P(e.get("bp"))
        a.WpP, a.WpTP = P(e.get("WpP")), P(e.get("WpTP"))
        a.N, a.K, a.tile0 = N, Kd, tile0
        tile0 += (N + 31) // 32
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    table = torch.from_numpy(raw).to(device)
    return table, len(entries), tile0


def prep_weights(table, n, tiles, plan=None):
    _emit(plan, L.lib().mmfm_prep_weights, (P(table), n, tiles), keep=(table,))


def rowgemm(x, w, y, R, N, K, *, ldx=None, ldw=None, ldy=None, bias=None, ln=False, eps=1e-5, xhat=None, rstd=None, residual=None,
            ldr=0, stream_out=False, ln_bwd=False, bwd_xhat=None, bwd_rstd=None, rotate=True, plan=None):
    d = L.RowGemmDesc()
    d.R, d.K, d.N = R, K, N
    d.x, d.ldx, d.w, d.ldw = P(x), K if ldx is None else ldx, P(w), K if ldw is None else ldw
    d.bias, d.ln, d.eps, d.xhat, d.rstd = P(bias), int(ln), eps, P(xhat), P(rstd)
    d.residual, d.ldr, d.y, d.ldy = P(residual), ldr, P(y), N if ldy is None else ldy
    d.stream_out, d.ln_bwd, d.bwd_xhat, d.bwd_rstd, d.rotate = int(stream_out), int(ln_bwd), P(bwd_xhat), P(bwd_rstd), int(rotate)
    _emit(plan, L.lib().mmfm_rowgemm, (C.byref(d),), keep=(d,))


def mlp_desc(R, *, x=None, ldx=256, eps=1e-5, w_up=None, b_up=None, w_down=None, b_down=None, drop=None, y=None, ldy=256, xhat=None,
             rstd=None, dy=None, lddy=256, w_down_t=None, w_up_t=None, t1=None, g=None, du=None, dx=None, lddx=256, rotate=True):
    d = L.MlpDesc()
    d.R, d.x, d.ldx, d.eps = R, P(x), ldx, eps
    d.w_up, d.b_up, d.w_down, d.b_down = P(w_up), P(b_up), P(w_down), P(b_down)
    d.drop = drop if drop is not None else L.NO_DROP
    d.y, d.ldy, d.xhat, d.rstd = P(y), ldy, P(xhat), P(rstd)
    d.dy, d.lddy, d.w_down_t, d.w_up_t = P(dy), lddy, P(w_down_t), P(w_up_t)
    d.t1, d.g, d.du, d.dx, d.lddx, d.rotate = P(t1), P(g), P(du), P(dx), lddx, int(rotate)
    return d


def mlp_fwd(desc, plan=None):
    _emit(plan, L.lib().mmfm_mlp_fwd, (C.byref(desc),), keep=(desc,))


def mlp_bwd(desc, plan=None):
    _emit(plan, L.lib().mmfm_mlp_bwd, (C.byref(desc),), keep=(desc,))


def ln_linear_grad_workspace(K, device):
    """Zeroed workspace of mmfm_ln_linear_grad (partials + self re-arming tickets)."""
    return torch.zeros(L.lib().mmfm_ln_linear_grad_workspace(K) // 4, dtype=torch.float32, device=device)


def ln_linear_grad(Gdb, W, gamma, beta, N, K, dW, dbias, dgamma, dbeta, ws, accumulate_ln=False, plan=None):
    _emit(plan, L.lib().mmfm_ln_linear_grad, (P(Gdb), P(W), P(gamma), P(beta), N, K, P(dW), P(dbias), P(dgamma), P(dbeta), int(accumulate_ln),
                                              P(ws), ws.numel() * 4), keep=(ws,))

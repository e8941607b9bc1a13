"""Transformer building blocks of the API mirror (reference: multi_modal/mm_utils.py:17-152).

These modules OWN PARAMETERS with the reference's names, shapes and initialisation order; they do
not carry the arithmetic.  On the training path `MultiModal.forward` hands the whole step to the
HIP engine (multi_modal_foundation_model_amd/engine.py); the stand-alone `forward` methods below
run the same C-ABI kernels op by op (inference only, no autograd) for callers that use a block in
isolation.  There is no torch fallback.
"""
import math

import torch
import torch.nn as nn

from multi_modal_foundation_model_amd import _lib as L
from multi_modal_foundation_model_amd import ops as K


def create_context_mask(context_forward, context_backward, max_F) -> torch.LongTensor:
    """[max_F, max_F] int64; with (0, -1) this is the causal mask 'key j <= query i' (mm_utils.py:17-28)."""
    if context_forward == -1 and context_backward == -1:
        return torch.ones(max_F, max_F, dtype=torch.int64)
    fwd = context_forward if context_forward >= 0 else max_F
    bwd = context_backward if context_backward >= 0 else max_F
    mask = torch.triu(torch.ones(max_F, max_F), diagonal=-fwd).to(torch.int64).transpose(0, 1)
    if bwd > 0:
        mask = mask & torch.triu(torch.ones(max_F, max_F), diagonal=-bwd).to(torch.int64)
    return mask


def _need_cuda(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the HIP kernels need device tensors; there is no CPU path in this package")


def _flat2d(x):
    return x.contiguous().view(-1, x.shape[-1])


def hip_linear(x, lin, act=0, pre_out=None, residual=None):
    """y = act(x @ W^T + b) through mmfm_gemm (fp32)."""
    _need_cuda(x, "hip_linear")
    x2 = _flat2d(x.float())
    M, Kd = x2.shape
    N = lin.weight.shape[0]
    y = torch.empty(M, N, device=x.device)
    K.gemm(x2, lin.weight.detach().float().contiguous(), y, M, N, Kd, lda=Kd, ldb=Kd, ldc=N,
           bias=None if lin.bias is None else lin.bias.detach().float().contiguous(), act=act, pre_out=pre_out,
           residual=residual, ldr=N if residual is not None else 0)
    return y.view(*x.shape[:-1], N)


def hip_layernorm(x, ln):
    _need_cuda(x, "hip_layernorm")
    x2 = _flat2d(x.float())
    R, H = x2.shape
    y, mean, rstd = torch.empty_like(x2), torch.empty(R, device=x.device), torch.empty(R, device=x.device)
    K.layernorm_fwd(x2, ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), y, mean, rstd, R, H, ln.eps)
    return y.view_as(x)


class ScaleNorm(nn.Module):
    """Declared for config parity (mm_utils.py:31-39); `use_scalenorm: true` has no HIP kernel."""

    def __init__(self, scale, eps=1e-5):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale))
        self.eps = eps

    def forward(self, x):
        raise NotImplementedError("ScaleNorm is not built for the MI355X path (mm.yaml: use_scalenorm=false)")


class MLP(nn.Module):
    def __init__(self, hidden_size, inter_size, act, use_bias, dropout):
        super().__init__()
        if act != "gelu":
            raise NotImplementedError(f"MLP act '{act}': only exact-erf gelu has a kernel")
        self.up_proj = nn.Linear(hidden_size, inter_size, bias=use_bias)
        self.down_proj = nn.Linear(inter_size, hidden_size, bias=use_bias)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        if self.training and self.dropout.p > 0:
            raise RuntimeError("stand-alone MLP.forward is inference-only; train through MultiModal.forward")
        return hip_linear(hip_linear(x, self.up_proj, act=L.ACT_GELU), self.down_proj)


class _AttentionBase(nn.Module):
    def __init__(self, idx, hidden_size, n_heads, use_bias, dropout):
        super().__init__()
        self.idx = idx
        self.hidden_size = hidden_size
        self.n_heads = n_heads
        assert self.hidden_size % self.n_heads == 0, "Hidden dim is not multiple of head size"
        self.head_size = self.hidden_size // self.n_heads
        self.query = nn.Linear(hidden_size, hidden_size, bias=use_bias)
        self.key = nn.Linear(hidden_size, hidden_size, bias=use_bias)
        self.value = nn.Linear(hidden_size, hidden_size, bias=use_bias)
        self.attn_dropout = dropout
        self.dropout = nn.Dropout(dropout)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=use_bias)

    def _attend(self, x, context, mask):
        """`mask`: the reference's [B, Lq, Lk] integer mask; it must be key padding, optionally | eye."""
        if self.training and self.attn_dropout > 0:
            raise RuntimeError("stand-alone attention forward is inference-only; train through MultiModal.forward")
        B, Lq, H = x.shape
        Lk = context.shape[1]
        mb = mask.bool()
        flags = 0
        if Lq == Lk:
            off_diag = mb & ~torch.eye(Lq, dtype=torch.bool, device=mb.device)
            keypad = off_diag.any(dim=1) | (mb.all(dim=1))
            rebuilt = keypad[:, None, :].expand(B, Lq, Lk)
            if not torch.equal(rebuilt, mb):
                flags = L.ATTN_DIAG
                rebuilt = rebuilt | torch.eye(Lq, dtype=torch.bool, device=mb.device)
        else:
            keypad = mb.any(dim=1)
            rebuilt = keypad[:, None, :].expand(B, Lq, Lk)
        if not torch.equal(rebuilt, mb):
            raise NotImplementedError("attention mask is not keypad(+diagonal): use MultiModal.forward (mask flags)")
        q, k, v = hip_linear(x, self.query), hip_linear(context, self.key), hip_linear(context, self.value)
        o = torch.empty(B * Lq, H, device=x.device)
        lse = torch.empty(B, self.n_heads, Lq, device=x.device)
        kp = keypad.to(torch.uint8).contiguous()
        desc = K.attn_desc(L.F32, B, self.n_heads, Lq, Lk, self.head_size, q.data_ptr(), k.data_ptr(), v.data_ptr(), H, H, H,
                           o.data_ptr(), H, lse, kp, None, flags, 1.0 / math.sqrt(self.head_size))
        K.attn_fwd(desc)
        return hip_linear(o.view(B, Lq, H), self.out_proj)


class Attention(_AttentionBase):
    def forward(self, x, mask):
        return self._attend(x, x, mask)


class CrossAttention(_AttentionBase):
    def forward(self, x, context, mask=None):
        return self._attend(x, context, mask)

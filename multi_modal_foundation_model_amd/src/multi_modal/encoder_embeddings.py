"""Encoder-side modality tokeniser and transformer block (reference:
multi_modal/encoder_embeddings.py:19-129).  Parameter holders with the reference's names and
creation order (token_embed, projection, mod_emb, pos_embed; ln1, attn, ln2, mlp, then the fixup
rescale).  The arithmetic of the training path runs in the HIP engine."""
import torch
import torch.nn as nn

from multi_modal.mm_utils import MLP, Attention, ScaleNorm, hip_layernorm, hip_linear
from multi_modal_foundation_model_amd import _lib as L
from multi_modal_foundation_model_amd import ops as K
from utils.config_utils import DictConfig

DEFAULT_CONFIG = "src/configs/multi_modal/mm.yaml"


class TokeniserLayer(nn.Module):
    """x = proj(softsign(tok(inputs)) * scale);  emb = mod_emb[modality] + pos_embed[timestamp]."""

    def __init__(self, hidden_size, n_channels, config: DictConfig):
        super().__init__()
        if config.act != "softsign":
            raise NotImplementedError(f"embedder act '{config.act}': only softsign has a kernel")
        self.bias = config.bias
        self.n_channels = n_channels
        self.input_dim = n_channels * config.mult
        self.token_embed = nn.Linear(n_channels, self.input_dim, bias=self.bias)
        self.projection = nn.Linear(self.input_dim, hidden_size)
        self.scale = hidden_size ** 0.5 if config.scale is None else config.scale
        self.mod_emb = nn.Embedding(config.n_modality, hidden_size)
        self.pos = config.pos
        if self.pos:
            self.pos_embed = nn.Embedding(config.max_F, hidden_size)
        else:
            raise NotImplementedError("embedder.pos=false is not built")
        self.dropout = nn.Dropout(config.dropout)

    def forward(self, d):
        """Stand-alone (inference) tokenisation through the HIP kernels; returns (x, emb)."""
        if self.training and self.dropout.p > 0:
            raise RuntimeError("stand-alone tokeniser forward is inference-only; train through MultiModal.forward")
        inputs, ts, mod = d["inputs"], d["inputs_timestamp"], int(d["inputs_modality"])
        B, T, _ = inputs.shape
        H = self.projection.weight.shape[0]
        a = hip_linear(inputs, self.token_embed, act=L.ACT_SOFTSIGN)
        if self.scale != 1:
            a = a * self.scale
        tok = hip_linear(a, self.projection).view(B * T, H)
        x, emb = torch.empty(B, T, H, device=inputs.device), torch.empty(B, T, H, device=inputs.device)
        keep = torch.ones(T, dtype=torch.uint8, device=inputs.device)
        K.stitch_fwd(tok, self.mod_emb.weight.detach()[mod].contiguous(), self.pos_embed.weight.detach().contiguous(),
                     ts.contiguous(), keep, x, emb, B, T, T, 0, H, self.pos_embed.weight.shape[0])
        return x - emb, emb


class EncoderEmbeddingLayer(TokeniserLayer):
    pass


class EncoderEmbedding(nn.Module):

    def __init__(self, n_channel, config: DictConfig, **kwargs):
        super().__init__()
        self.hidden_size = config.transformer.hidden_size
        self.n_layers = config.transformer.n_layers
        self.max_F = config.embedder.max_F
        self.n_channel = n_channel
        self.embedder = EncoderEmbeddingLayer(self.hidden_size, self.n_channel, config.embedder)

    def forward(self, d):
        d["x"], d["emb"] = self.embedder(d)
        return d


def fixup_rescale(block: nn.Module, n_layers: int):
    """`*_proj.weight` x 0.67 n^-1/4, `value.weight` x 0.67 n^-1/4 sqrt(2)  (same operation order as the
    reference so the result is bit-identical, encoder_embeddings.py:118-129)."""
    c = 0.67 * n_layers ** (-1.0 / 4.0)
    with torch.no_grad():
        for name, prm in block.named_parameters():
            if name.endswith("_proj.weight"):
                prm.copy_(c * prm)
            elif name.endswith("value.weight"):
                prm.copy_(c * (prm * (2 ** 0.5)))


def make_norm(config):
    return ScaleNorm(config.hidden_size ** 0.5) if config.use_scalenorm else nn.LayerNorm(config.hidden_size)


class EncoderLayer(nn.Module):

    def __init__(self, idx, config: DictConfig):
        super().__init__()
        self.idx = idx
        self.ln1 = make_norm(config)
        self.attn = Attention(idx, config.hidden_size, config.n_heads, config.attention_bias, config.dropout)
        self.ln2 = make_norm(config)
        self.mlp = MLP(config.hidden_size, config.inter_size, config.act, config.mlp_bias, config.dropout)
        if config.fixup_init:
            self.fixup_initialization(config.n_layers)

    def forward(self, x, mask):
        x = x + self.attn(hip_layernorm(x, self.ln1), mask)
        return x + self.mlp(hip_layernorm(x, self.ln2))

    def fixup_initialization(self, n_layers):
        fixup_rescale(self, n_layers)

"""Decoder-side tokeniser, output head and transformer block (reference:
multi_modal/decoder_embeddings.py:19-160).  Parameter holders; see encoder_embeddings.py."""
import torch.nn as nn

from multi_modal.encoder_embeddings import TokeniserLayer, fixup_rescale, make_norm
from multi_modal.mm_utils import MLP, Attention, CrossAttention, hip_layernorm, hip_linear
from utils.config_utils import DictConfig

DEFAULT_CONFIG = "src/configs/multi_modal/mm.yaml"


class DecoderEmbeddingLayer(TokeniserLayer):
    """Same arithmetic as the encoder tokeniser on the same (unmasked) inputs, separate weights
    (decoder_embeddings.py:43-61)."""


class DecoderEmbedding(nn.Module):

    def __init__(self, n_channel, output_channel, config: DictConfig, **kwargs):
        super().__init__()
        self.hidden_size = config.transformer.hidden_size
        self.n_layers = config.transformer.n_layers
        self.max_F = config.embedder.max_F
        self.n_channel = n_channel
        self.output_channel = output_channel
        self.embedder = DecoderEmbeddingLayer(self.hidden_size, self.n_channel, config.embedder)
        self.out = nn.Linear(self.hidden_size, self.output_channel)

    def forward_embed(self, d):
        d["x"], d["emb"] = self.embedder(d)
        d["gt"] = d["targets"]
        return d

    def out_proj(self, mod_idx, d, y, decoder_mod_mask, n_mod):
        B = y.shape[0]
        d["preds"] = hip_linear(y[decoder_mod_mask == mod_idx], self.out).reshape((B, -1, self.output_channel))
        return d


class DecoderLayer(nn.Module):

    def __init__(self, idx, config: DictConfig):
        super().__init__()
        self.idx = idx
        self.ln1 = make_norm(config)
        self.attn = Attention(idx, config.hidden_size, config.n_heads, config.attention_bias, config.dropout)
        self.cross_attn = CrossAttention(idx, config.hidden_size, config.n_heads, config.attention_bias, config.dropout)
        self.query_norm = make_norm(config)
        self.context_norm = make_norm(config)
        self.ln2 = make_norm(config)
        self.mlp = MLP(config.hidden_size, config.inter_size, config.act, config.mlp_bias, config.dropout)
        if config.fixup_init:
            self.fixup_initialization(config.n_layers)

    def forward(self, x, context, sa_mask=None, xa_mask=None):
        x = x + self.attn(hip_layernorm(x, self.ln1), sa_mask)
        x = x + self.cross_attn(hip_layernorm(x, self.query_norm), hip_layernorm(context, self.context_norm), xa_mask)
        return x + self.mlp(hip_layernorm(x, self.ln2))

    def fixup_initialization(self, n_layers):
        fixup_rescale(self, n_layers)

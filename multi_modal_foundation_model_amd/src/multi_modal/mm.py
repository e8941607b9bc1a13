"""`MultiModal` — the reference's encoder-decoder masked model class (multi_modal/mm.py:33-308)
with the same constructor, attributes, parameter names and `forward(mod_dict)` contract, whose
forward/backward run as hand-written HIP kernels (multi_modal_foundation_model_amd/engine.py).

Behavioural notes kept from the reference on purpose:
  * `forward` mutates `mod_dict` in place (inputs_mask, targets_mask, *_attn_mask, gt, preds);
  * tokens are zeroed at the positions where SAMPLE 0 is masked, for every sample (mm.py:147-149);
  * the loss is sum(mod_loss)/sum(n_examples) and is NaN when nothing is masked (mm.py:237);
  * `masking_mode` (mask_type: input) fails exactly like upstream (`mask` is never bound, mm.py:256-272).
"""
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from models.masker import Masker
from models.model_output import ModelOutput
from multi_modal.decoder_embeddings import DecoderLayer
from multi_modal.encoder_embeddings import EncoderLayer
from multi_modal.mm_utils import create_context_mask  # noqa: F401  (re-exported like the reference)
from multi_modal_foundation_model_amd.engine import Engine, EngineConfig
from utils.config_utils import DictConfig

DEFAULT_CONFIG = "src/configs/multi_modal/mm.yaml"


@dataclass
class MultiModalOutput(ModelOutput):
    loss: Optional[torch.FloatTensor] = None
    mod_loss: Optional[Dict[str, torch.FloatTensor]] = None
    mod_n_examples: Optional[Dict[str, torch.LongTensor]] = None
    mod_preds: Optional[Dict[str, torch.FloatTensor]] = None
    mod_targets: Optional[Dict[str, torch.FloatTensor]] = None


def _loss_kind(spec) -> int:
    """loss_mod entry -> mmfm_masked_loss kind.  Entries are strings here; the reference's entries are
    nn.PoissonNLLLoss(reduction="none", log_input=True) / nn.MSELoss(reduction="none") (mm.py:79-82), accepted too, so
    `model.loss_mod['lfp'] = nn.MSELoss(reduction='none')` adds a modality exactly as it would upstream."""
    name = (spec if isinstance(spec, str) else type(spec).__name__).lower()
    if "poisson" in name:
        if not isinstance(spec, str) and not getattr(spec, "log_input", True):
            raise NotImplementedError("PoissonNLLLoss(log_input=False) has no HIP kernel")
        return 0
    if "mse" in name:
        return 1
    raise NotImplementedError(f"loss {spec!r}: only PoissonNLL(log_input) and MSE are built")


class MultiModal(nn.Module):

    def __init__(self, encoder_embeddings: Dict[str, nn.Module], decoder_embeddings: Dict[str, nn.Module],
                 avail_mod: List, config: DictConfig, share_modality_embeddings: bool = True, **kwargs):
        super().__init__()
        self.avail_mod = avail_mod
        self.mod_to_indx = {r: i for i, r in enumerate(self.avail_mod)}
        self.decoder_sep_mask = config.decoder.decoder_sep_mask
        self.decoder_causal_mask = config.decoder.decoder_causal_mask
        self.n_enc_layers = config.encoder.transformer.n_layers
        self.n_dec_layers = config.decoder.transformer.n_layers
        self.hidden_size = config.encoder.transformer.hidden_size
        self.max_F = config.encoder.embedder.max_F
        self.context_forward = config.context.forward
        self.context_backward = config.context.backward

        self.encoder_modalities = set(encoder_embeddings.keys())
        self.encoder_embeddings = nn.ModuleDict(encoder_embeddings)
        self.decoder_modalities = set(decoder_embeddings.keys())
        self.decoder_embeddings = nn.ModuleDict(decoder_embeddings)
        if share_modality_embeddings:
            self.share_modality_embeddings()
        elif self.encoder_modalities & self.decoder_modalities:
            raise NotImplementedError("share_modality_embeddings=False is not built (the entry script passes True)")

        self.mask = config.masker.force_active
        if self.mask:
            assert config.masker.mode in ['temporal'], "Only token-wise masking is allowed for multi-modal model for now."
            self.masker = Masker(config.masker)

        self.encoder = nn.ModuleList([EncoderLayer(i, config.encoder.transformer) for i in range(self.n_enc_layers)])
        self.encoder_norm = nn.LayerNorm(self.hidden_size)
        self.decoder_proj_context = nn.Linear(self.hidden_size, self.hidden_size)
        self.decoder = nn.ModuleList([DecoderLayer(i, config.decoder.transformer) for i in range(self.n_dec_layers)])
        self.decoder_norm = nn.LayerNorm(self.hidden_size)
        # loss per modality (mm.py:79-82): 'ap' PoissonNLL(log_input), 'behavior' MSE — computed by mmfm_masked_loss_*
        self.loss_mod = {"ap": "poisson_nll_log_input", "behavior": "mse"}

        self._model_config = config
        self._engine: Optional[Engine] = None
        self._sentinels = None
        # "fp32": parity mode (fp32 MFMA); "bf16": throughput mode (bf16 storage/MFMA, fp32 accumulate + master weights)
        self.compute_dtype = os.environ.get("MMFM_DTYPE", "fp32")
        self.engine_seed = 0

    def share_modality_embeddings(self):
        for mod in self.encoder_modalities & self.decoder_modalities:
            self.decoder_embeddings[mod].embedder.mod_emb = self.encoder_embeddings[mod].embedder.mod_emb

    # ------------------------------------------------------------------ engine plumbing
    def engine(self) -> Engine:
        # fast path (every forward): the cached engine is still valid if a few sentinel parameters are still
        # views of its flat buffer on the same device (.to()/load_state_dict(assign=True) replace them all)
        eng = self._engine
        if eng is not None and eng.dtype == self.compute_dtype and self._sentinels and \
                all(eng.owns_one(p) for p in self._sentinels):
            return eng
        named = dict(self.named_parameters())
        plist = list(named.values())
        self._sentinels = [plist[0], plist[len(plist) // 2], plist[-1]]
        dev = next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError("MultiModal runs on an MI355X through libmmfm_hip.so; move the model to the GPU first "
                               "(there is deliberately no CPU fallback; the CPU restatement is oracle/, test-only)")
        if self._engine is None or self._engine.device != dev or self._engine.dtype != self.compute_dtype:
            mods = [(m, self.encoder_embeddings[m].n_channel) for m in self.avail_mod]
            for m, n in mods:
                if self.decoder_embeddings[m].n_channel != n or self.decoder_embeddings[m].output_channel != n:
                    raise NotImplementedError("encoder/decoder channel counts differ")
                if m not in self.loss_mod:
                    raise Exception("Modality not implemented yet.")
            cfg = EngineConfig.from_model_config(self._model_config, mods)
            cfg.loss_kind = {m: _loss_kind(self.loss_mod[m]) for m, _ in mods}
            self._engine = Engine(cfg, dev, dtype=self.compute_dtype, seed=self.engine_seed)
            self._engine.adopt(named)
        elif not self._engine.owns(named):
            self._engine.adopt(named)        # parameters were replaced (.to(), load_state_dict(assign=True), ...)
        return self._engine

    def __getstate__(self):                   # torch.save(model) (trainer/base.py:302-308): parameters own their data again
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_sentinels"] = None
        return state

    # ------------------------------------------------------------------ forward
    def forward(self, mod_dict: Dict[str, Dict[str, Any]]) -> MultiModalOutput:
        mods = list(mod_dict.keys())
        if mods != list(self.avail_mod):
            raise Exception(f"mod_dict modalities {mods} != avail_mod {list(self.avail_mod)}")
        masks = []
        for mod in mods:
            d = mod_dict[mod]
            if mod == 'behavior' and d['inputs'].dim() == 2:
                d['inputs'] = d['inputs'].unsqueeze(-1)
                d['targets'] = d['targets'].unsqueeze(-1)
            regions = d['inputs_regions'] if mod == 'ap' else None
            if d['masking_mode']:
                self.masker.mode = d['masking_mode']
                d['inputs'], d['spike_mask'] = self.masker(d['inputs'].clone(), regions)
                raise UnboundLocalError("local variable 'mask' referenced before assignment "
                                        "(upstream behaviour of mask_type='input', mm.py:256-272)")
            if d['eval_mask'] is None:
                # the corrupted spikes are discarded here (mm.py:267), so the trainer's token-mask-only switch applies
                _, mask = self.masker(d['inputs'].clone(), regions, token_mask_only=bool(self.masker.token_mask_only))
            else:
                mask = d['eval_mask']
            mask = mask[:, :, 0] & d['inputs_attn_mask']
            d['inputs_mask'] = d['targets_mask'] = mask
            d['encoder_attn_mask'] = d['decoder_attn_mask'] = d['inputs_attn_mask']
            masks.append(mask)
        first = mod_dict[mods[0]]
        ts, attn = first['inputs_timestamp'], first['inputs_attn_mask']
        for mod in mods[1:]:
            d = mod_dict[mod]
            for key, ref in (('inputs_timestamp', ts), ('inputs_attn_mask', attn)):
                if d[key] is not ref and not torch.equal(d[key], ref):
                    raise NotImplementedError(f"per-modality {key} differ: the stitched sequence assumes shared bins")
        B, T, _ = first['inputs'].shape
        eng = self.engine()
        out = eng.forward(B, T, [mod_dict[m]['inputs'] for m in mods], [mod_dict[m]['targets'] for m in mods], masks, ts, attn,
                          training=self.training, anchor=self.decoder_norm.weight)
        mod_loss, mod_n, preds, targets = {}, {}, {}, {}
        for i, mod in enumerate(mods):
            mod_loss[mod], mod_n[mod] = out["mod_loss"][i], out["mod_n"][i]
            # bf16 engine: the fp32 copy of the predictions (68 M elements for 'ap' at B = 1024: a 410 MB cast kernel per step) is made
            # where somebody reads them - evaluation; in training mode (the trainer's train_epoch only reads the loss) mod_preds
            # carries the engine's bf16 predictions as they are
            p_ = out["preds"][i]
            preds[mod] = p_ if (p_.dtype == torch.float32 or self.training) else p_.float()
            targets[mod] = mod_dict[mod]['targets']
            mod_dict[mod]['gt'], mod_dict[mod]['preds'] = targets[mod], preds[mod]
        return MultiModalOutput(loss=out["loss"], mod_loss=mod_loss, mod_n_examples=mod_n, mod_preds=preds, mod_targets=targets)

"""The train-step engine: MultiModal forward / backward as a pre-bound plan of HIP kernel launches.

Reference path: `MultiModal.forward` (src/multi_modal/mm.py:242-308) + autograd of it
(`loss.backward()`, src/trainer/base.py:194-195).  Design (MI355X-first, not a translation):

* all parameters live in ONE flat fp32 buffer (`P`), gradients in a second (`G`); the nn.Module
  parameters of the API mirror are views into it.  The layout is forward order so that the
  gradient ranges complete back-to-front during backward: the DDP wrapper all-reduces contiguous
  buckets of `G` as soon as backward passes their start, overlapped with the rest of backward.
  Q/K/V (and cross-attention K/V) weights are adjacent, so one GEMM does the fused projection.
* activations/workspaces are allocated once per batch shape; a step is a fixed list of
  (C function, bound arguments) pairs -> no per-step allocation, marshalling or host sync, and
  the list can be captured into a hipGraph (`Engine.capture`).
* backward is written by hand (no autograd graph): LayerNorm backward fuses the residual-gradient
  add, dX GEMMs fuse the activation derivative, dW GEMMs are split-K over the token dimension with
  a deterministic slab reduction, dropout masks are regenerated from a counter RNG.
* there is no CPU / eager fallback: everything below calls libmmfm_hip.so.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import ops as K

LOSS_KIND = {"ap": 0, "behavior": 1}       # mm.py:79-82: PoissonNLL(log_input) / MSE


@dataclass
class EngineConfig:
    hidden: int
    heads: int
    inter: int
    n_enc: int
    n_dec: int
    max_F: int
    mult: int
    n_modality: int
    embed_scale: float
    embed_dropout: float
    dropout: float
    sep_mask: bool
    causal_mask: bool
    mods: List[Tuple[str, int]]                 # (name, channels) in avail_mod order
    loss_kind: Dict[str, int] = field(default_factory=lambda: dict(LOSS_KIND))

    @staticmethod
    def from_model_config(mc, mods) -> "EngineConfig":
        et, ee = mc["encoder"]["transformer"], mc["encoder"]["embedder"]
        dtf = mc["decoder"]["transformer"]
        for k in ("hidden_size", "n_heads", "inter_size", "dropout"):
            if et[k] != dtf[k]:
                raise ValueError(f"encoder/decoder transformer.{k} differ ({et[k]} vs {dtf[k]}): not supported")
        if et["use_scalenorm"] or dtf["use_scalenorm"]:
            raise NotImplementedError("use_scalenorm=true has no HIP kernel (mm.yaml default is false)")
        if et["act"] != "gelu" or ee["act"] != "softsign":
            raise NotImplementedError("only act=gelu (transformer) / softsign (embedder) are built")
        scale = et["hidden_size"] ** 0.5 if ee["scale"] is None else ee["scale"]
        return EngineConfig(hidden=et["hidden_size"], heads=et["n_heads"], inter=et["inter_size"],
                            n_enc=et["n_layers"], n_dec=dtf["n_layers"], max_F=ee["max_F"], mult=ee["mult"],
                            n_modality=ee["n_modality"], embed_scale=float(scale), embed_dropout=ee["dropout"],
                            dropout=et["dropout"], sep_mask=bool(mc["decoder"]["decoder_sep_mask"]),
                            causal_mask=bool(mc["decoder"]["decoder_causal_mask"]), mods=list(mods))


def _align(n, a=8):
    return (n + a - 1) // a * a


class ParamLayout:
    """name -> (offset, shape) in the flat buffer; `groups` are the DDP buckets' atoms."""

    def __init__(self, cfg: EngineConfig):
        H, I = cfg.hidden, cfg.inter
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.alias: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.segments: List[Tuple[str, int, int]] = []       # (segment name, start, end) in forward order
        self.n = 0

        def add(name, shape):
            self.entries[name] = (self.n, tuple(shape))
            self.n += int(math.prod(shape))

        def pad():
            self.n = _align(self.n)

        def seg_begin():
            pad()
            return self.n

        def lin(prefix, o, i):
            pad(); add(prefix + ".weight", (o, i)); pad(); add(prefix + ".bias", (o,))

        def ln(prefix):
            pad(); add(prefix + ".weight", (H,)); pad(); add(prefix + ".bias", (H,))

        def fused(prefix, names, alias):
            pad()
            start = self.n
            for nm in names:
                add(f"{prefix}.{nm}.weight", (H, H))
            self.alias[f"{prefix}.{alias}.weight"] = (start, (len(names) * H, H))
            pad()
            start = self.n
            for nm in names:
                add(f"{prefix}.{nm}.bias", (H,))
            self.alias[f"{prefix}.{alias}.bias"] = (start, (len(names) * H,))

        s = seg_begin()
        for side in ("encoder", "decoder"):
            for mod, n in cfg.mods:
                p = f"{side}_embeddings.{mod}.embedder"
                lin(p + ".token_embed", n * cfg.mult, n)
                lin(p + ".projection", H, n * cfg.mult)
                if side == "encoder":       # decoder's mod_emb IS this tensor (mm.py:84-87)
                    pad(); add(p + ".mod_emb.weight", (cfg.n_modality, H))
                pad(); add(p + ".pos_embed.weight", (cfg.max_F, H))
        pad()
        self.segments.append(("embed", s, self.n))
        for i in range(cfg.n_enc):
            s = seg_begin()
            p = f"encoder.{i}"
            ln(p + ".ln1"); fused(p + ".attn", ("query", "key", "value"), "qkv"); lin(p + ".attn.out_proj", H, H)
            ln(p + ".ln2"); lin(p + ".mlp.up_proj", I, H); lin(p + ".mlp.down_proj", H, I)
            pad()
            self.segments.append((p, s, self.n))
        s = seg_begin()
        ln("encoder_norm"); lin("decoder_proj_context", H, H)
        pad()
        self.segments.append(("bridge", s, self.n))
        for i in range(cfg.n_dec):
            s = seg_begin()
            p = f"decoder.{i}"
            ln(p + ".ln1"); fused(p + ".attn", ("query", "key", "value"), "qkv"); lin(p + ".attn.out_proj", H, H)
            ln(p + ".query_norm"); ln(p + ".context_norm")
            lin(p + ".cross_attn.query", H, H); fused(p + ".cross_attn", ("key", "value"), "kv")
            lin(p + ".cross_attn.out_proj", H, H)
            ln(p + ".ln2"); lin(p + ".mlp.up_proj", I, H); lin(p + ".mlp.down_proj", H, I)
            pad()
            self.segments.append((p, s, self.n))
        s = seg_begin()
        ln("decoder_norm")
        for mod, n in cfg.mods:
            lin(f"decoder_embeddings.{mod}.out", n, H)
        pad()
        self.segments.append(("head", s, self.n))
        self.n = _align(self.n, 64)

    def view(self, flat, name):
        off, shape = self.entries[name] if name in self.entries else self.alias[name]
        return flat[off: off + int(math.prod(shape))].view(shape)


class _StepFn(torch.autograd.Function):
    """Connects the engine to `loss.backward()` (trainer/base.py:194-195)."""

    @staticmethod
    def forward(ctx, anchor, engine, token):
        ctx.engine, ctx.token = engine, token
        return engine.b["loss"].clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        ctx.engine.backward(grad_out, ctx.token)
        return None, None, None


class Engine:
    def __init__(self, cfg: EngineConfig, device, dtype: str = "fp32", seed: int = 0):
        if cfg.hidden % cfg.heads:
            raise ValueError("Hidden dim is not multiple of head size")
        if dtype not in ("fp32", "bf16"):
            raise ValueError(dtype)
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X engine needs a CUDA/HIP device; there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        L.check(L.lib().mmfm_device_check(self.device.index or 0), "mmfm_device_check")
        self.dtype = dtype
        self.adt = torch.float32 if dtype == "fp32" else torch.bfloat16
        self.code = L.F32 if dtype == "fp32" else L.BF16
        self.layout = ParamLayout(cfg)
        n = self.layout.n
        self.P = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.G = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.Pw = self.P if dtype == "fp32" else torch.zeros(n, dtype=torch.bfloat16, device=self.device)
        self.rng = torch.zeros(2, dtype=torch.int32, device=self.device)
        K.rng_seed(self.rng, seed)
        self.params: Dict[str, torch.nn.Parameter] = {}
        self.plans: Dict[Tuple, dict] = {}
        # Buffers are owned per batch shape (B, T): a plan (and its captured hipGraphs) holds raw device pointers, so a
        # buffer must never be re-allocated while a plan that names it is alive.  `self.b` is the pool of the shape in
        # use; pools (with their plans) are evicted least-recently-used beyond MMFM_MAX_SHAPES.
        self._pools: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}
        self._pool_lru: List[Tuple[int, int]] = []
        self.max_shapes = max(1, int(os.environ.get("MMFM_MAX_SHAPES", "4")))
        self._shared: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = self._shared
        self._shape = None
        self._token = 0
        self._fwd_token = -1
        self._sites: Dict[str, int] = {}
        self.grad_ready_hooks = []       # DDP: callables(segment_name) fired as backward completes a segment of G
        self.backward_done_hooks = []    # DDP: wait for the collectives (stream-side) before anyone reads G
        # hipGraph replay of the step plan: the plan neither allocates nor synchronises, so after one eager
        # (warm-up) run per batch shape it is captured once and replayed.  MMFM_GRAPH=0 disables.
        self.use_graphs = os.environ.get("MMFM_GRAPH", "1") != "0"
        self._shared["loss"] = torch.zeros(1, device=self.device)
        self._shared["inv_n"] = torch.zeros(1, device=self.device)
        self._shared["gout"] = torch.ones(1, device=self.device)

    # ------------------------------------------------------------------ parameters
    def adopt(self, named_params: Dict[str, torch.nn.Parameter]):
        """Copy the module's parameters into the flat buffer and re-point them at views of it."""
        missing = set(self.layout.entries) - set(named_params)
        extra = set(named_params) - set(self.layout.entries)
        if missing or extra:
            raise KeyError(f"parameter set mismatch: missing {sorted(missing)[:4]}, unexpected {sorted(extra)[:4]}")
        with torch.no_grad():
            for name, p in named_params.items():
                v = self.layout.view(self.P, name)
                if tuple(p.shape) != tuple(v.shape):
                    raise ValueError(f"{name}: shape {tuple(p.shape)} != {tuple(v.shape)}")
                v.copy_(p.data.to(self.device, torch.float32))
                p.data = v
                p.grad = None
                self.params[name] = p
        self.refresh_weights()

    def owns(self, named_params) -> bool:
        """True while the module's parameters are still views of our flat buffer (e.g. `.to()` breaks it)."""
        lo, hi = self.P.data_ptr(), self.P.data_ptr() + self.P.numel() * 4
        return all(lo <= p.data_ptr() < hi for p in named_params.values())

    def owns_one(self, p) -> bool:
        lo = self.P.data_ptr()
        return p.device == self.P.device and lo <= p.data_ptr() < lo + self.P.numel() * 4

    def refresh_weights(self):
        """bf16 mode: refresh the bf16 weight copy from the fp32 master (the fused AdamW does it itself)."""
        if self.dtype == "bf16":
            K.cast_bf16(self.P, self.Pw, self.P.numel())

    def W(self, name):
        return self.layout.view(self.Pw, name)

    def Pf(self, name):          # fp32 master view (LayerNorm affine, biases, embedding tables)
        return self.layout.view(self.P, name)

    def Gv(self, name):
        return self.layout.view(self.G, name)

    # ------------------------------------------------------------------ buffers
    def _select_pool(self, B, T):
        """Make the buffer pool of batch shape (B, T) current (creating it, and evicting the least recently used
        shape - pool, plans and graphs together - beyond `max_shapes`)."""
        key = (B, T)
        pool = self._pools.get(key)
        if pool is None:
            while len(self._pools) >= self.max_shapes:
                old = self._pool_lru.pop(0)
                torch.cuda.synchronize(self.device)          # nothing may still be replaying the evicted graphs
                for pk in [k for k in self.plans if (k[0], k[1]) == old]:
                    del self.plans[pk]
                del self._pools[old]
            pool = dict(self._shared)
            self._pools[key] = pool
        if key in self._pool_lru:
            self._pool_lru.remove(key)
        self._pool_lru.append(key)
        self.b = pool
        return pool

    def _buf(self, name, shape, dtype=None, zero=False):
        dtype = self.adt if dtype is None else dtype
        t = self.b.get(name)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self.b[name] = t
        elif tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            raise RuntimeError(f"engine buffer {name}: {tuple(t.shape)}/{t.dtype} re-requested as {tuple(shape)}/{dtype} "
                               "inside one batch-shape pool (plans hold raw pointers; buffers are never re-allocated)")
        return t

    def _site(self, key):
        if key not in self._sites:
            self._sites[key] = len(self._sites) + 1
        return self._sites[key]

    def _drop(self, key, p):
        return K.dropout(self.rng, self._site(key), p) if p > 0 else None

    # ------------------------------------------------------------------ row-owner fused path (bf16, width 256 / 512)
    def _fused_mask(self, R):
        """Which op groups run as row-owner fused kernels (csrc/rowchain.h): bit 0 ln1+qkv, bit 1 the other LayerNorm-fed
        linears (cross-attention query / key-value, decoder_proj_context), bit 2 the MLP block, bit 3 attention out_proj.
        MMFM_FUSED overrides (0 = the un-fused kernels of round 1)."""
        c = self.cfg
        if self.dtype != "bf16" or c.hidden != 256 or c.inter != 512 or (R + 128) * 1024 * 2 >= 2 ** 31:
            return 0
        if R < 12288 and "MMFM_FUSED" not in os.environ:
            # a row-owner pass is 128 rows: below ~100 passes per launch the grid cannot fill 256 CUs.  Since round 4 the forward-type linears
            # split N into column blocks there (rowgemm.hip), which makes the LayerNorm-fed linears and out_proj worth fusing at the reference's
            # batch of 16 too; the MLP kernels (chained products, no column split) stay off.  ms/step un-fused / 11 / 15: B=16 4.45 / 4.08 / 4.19,
            # B=32 4.85 / 4.71 / -, B=64 5.46 / 5.50 / 5.48 (R = 12,800: the default 15)
            return 11
        return int(os.environ.get("MMFM_FUSED", "15")) & 15        # default: everything fused, the fastest end to end (35.6 vs 36.3 ms)

    # ------------------------------------------------------------------ bf16 transposes for the compute-bound dX products
    def _w_transposed(self, wname, N, Kd, Mr):
        """W^T [Kd, N] (bf16) of an nn.Linear weight [N, Kd] when its dX = dY[Mr, N] . W belongs to the 256-tile GEMM (csrc/gemm_big.hip:
        reduction N a multiple of 64 and >= 512; d_model-512 configurations and any layer this wide): that kernel wants both operands
        reduction-contiguous.  The views are refreshed by ONE mmfm_prep_weights launch at the top of every training step."""
        if os.environ.get("MMFM_GEMM_BIG", "1") == "0" or N < 512 or N % 64 or Kd % 8 or Kd < 128 or Mr < 1024:
            return None
        reg = self.__dict__.setdefault("_wt", dict(views={}, entries=[], table=None))
        if wname not in reg["views"]:
            t = torch.zeros(Kd, N, dtype=torch.bfloat16, device=self.device)
            reg["views"][wname] = t
            reg["entries"].append(dict(W=self.Pf(wname + ".weight"), WpT=t))
            reg["table"] = None
        return reg["views"][wname]

    def _wt_table(self):
        reg = self._wt
        if reg["table"] is None:
            table, n, tiles = K.prep_table(reg["entries"], self.device)
            reg["table"] = dict(table=table, n=n, tiles=tiles)
        return reg["table"]

    def _build_prep(self):
        """Prepared weights of the fused path: per LayerNorm-fed linear Wp = bf16(W * gamma), WpT, bp = b + W beta
        (mmfm_prep_weights); per plain linear only the bf16 transpose (the dX products read K-contiguous rows)."""
        if getattr(self, "_prep", None) is not None:
            return self._prep
        c = self.cfg
        sites = []
        for i in range(c.n_enc):
            p = f"encoder.{i}"
            sites += [(p + ".attn.qkv", p + ".ln1"), (p + ".attn.out_proj", None), (p + ".mlp.up_proj", p + ".ln2"), (p + ".mlp.down_proj", None)]
        sites.append(("decoder_proj_context", "encoder_norm"))
        for i in range(c.n_dec):
            p = f"decoder.{i}"
            sites += [(p + ".attn.qkv", p + ".ln1"), (p + ".attn.out_proj", None), (p + ".cross_attn.query", p + ".query_norm"),
                      (p + ".cross_attn.kv", p + ".context_norm"), (p + ".cross_attn.out_proj", None), (p + ".mlp.up_proj", p + ".ln2"),
                      (p + ".mlp.down_proj", None)]
        nW = sum(self.Pf(w + ".weight").numel() for w, _ in sites)
        nWp = sum(self.Pf(w + ".weight").numel() for w, ln in sites if ln)
        nb = sum(self.Pf(w + ".bias").numel() for w, ln in sites if ln)
        # unit-permuted copies for the MLP kernels' LDS-DMA weight ring (include/mmfm.h: mmfm_prep_entry.WpP / WpTP)
        nPm = sum(self.Pf(w + ".weight").numel() for w, _ in sites if w.endswith(".mlp.up_proj") or w.endswith(".mlp.down_proj"))
        WpT = torch.zeros(nW + 64, dtype=torch.bfloat16, device=self.device)
        Wp = torch.zeros(nWp + 64, dtype=torch.bfloat16, device=self.device)
        Wpm = torch.zeros(nPm + 64, dtype=torch.bfloat16, device=self.device)
        bp = torch.zeros(nb + 64, dtype=torch.float32, device=self.device)
        views, entries, oT, oW, ob, oP = {}, [], 0, 0, 0, 0
        for w, ln in sites:
            Wm = self.Pf(w + ".weight")
            N, Kd = Wm.shape
            e = dict(W=Wm, WpT=WpT[oT:oT + N * Kd].view(Kd, N))
            oT += N * Kd
            v = dict(WpT=e["WpT"])
            if w.endswith(".mlp.up_proj"):          # backward: d(x_hat) += W_up^T[:, tile] . du, du an accumulator tile
                e["WpTP"] = v["WpTP"] = Wpm[oP:oP + N * Kd].view(Kd, N)
                oP += N * Kd
            elif w.endswith(".mlp.down_proj"):      # forward: y += W_down[:, tile] . g, g an accumulator tile
                e["WpP"] = v["WpP"] = Wpm[oP:oP + N * Kd].view(N, Kd)
                oP += N * Kd
            if ln:
                e.update(gamma=self.Pf(ln + ".weight"), beta=self.Pf(ln + ".bias"), bias=self.Pf(w + ".bias"),
                         Wp=Wp[oW:oW + N * Kd].view(N, Kd), bp=bp[ob:ob + N])
                oW += N * Kd
                ob += N
                v.update(Wp=e["Wp"], bp=e["bp"])
            views[w] = v
            entries.append(e)
        table, n, tiles = K.prep_table(entries, self.device)
        self._prep = dict(table=table, n=n, tiles=tiles, v=views, keep=(WpT, Wp, Wpm, bp, entries))
        return self._prep

    # ------------------------------------------------------------------ plan construction
    def _dw_split(self, M, N, R, ldn=None):
        """Split-K factor for a dW GEMM ([M,N] output, reduction over R tokens).
        bf16 shapes the streaming kernel takes (csrc/gemm_dw.hip: 16-B aligned rows of dY [R, M] and X [R, ldn]): ONE (tile, K-slab) item per CU - the slab
        traffic is items x 64 KB, so fewer, longer items beat filling the chip three times over.
        Otherwise (fp32 parity path, the 668-wide tokeniser shapes): tiles x splits fills the persistent grid of the 128-tile
        kernel (256 CUs x 3 workgroups) exactly once - measured against 512 items at B = 1024: qkv dW 182 -> 155 us, token-embed
        dW 639 -> 508 us - capped at 128 slabs (the slab reduction costs S x M x N x 4 bytes)."""
        tiles = -(-M // 128) * -(-N // 128)
        stream = self.code == L.BF16 and M % 8 == 0 and (ldn or N) % 8 == 0 and os.environ.get("MMFM_GEMM_DW", "1") != "0"
        if stream:
            tiles = L.lib().mmfm_gemm_dw_tiles(M, N, R)
        if R <= 8192:                     # launch-bound regime (reference batch 16 -> R = 3200): <= 15 slabs = one-stage reduce
            S = max(1, min(R // 256, 15))
        elif stream:
            S = max(1, min(R // 512, 256 // tiles))
        else:
            S = max(1, min(R // 512, max(1, 768 // tiles), 128))
        kchunk = _align(-(-R // S), 64)          # multiple of both kernels' BK (32 fp32, 64 bf16)
        return -(-R // kchunk), kchunk

    def _plan(self, B, T, training, grad=True):
        """grad = False: a forward-only plan (evaluation under no_grad) that skips the tensors saved for the backward."""
        key = (B, T, bool(training), bool(grad))
        self._select_pool(B, T)
        if key in self.plans:
            return self.plans[key]
        c = self.cfg
        H, I, heads = c.hidden, c.inter, c.heads
        dh = H // heads
        M = len(c.mods)
        Lq = M * T
        R, BT = B * Lq, B * T
        dp = c.dropout if training else 0.0
        dpe = c.embed_dropout if training else 0.0
        f32, i64, u8 = torch.float32, torch.int64, torch.uint8
        buf = self._buf
        fwd, bwd_tail = [], []
        code = self.code

        # ---- static inputs
        for m, (mod, n) in enumerate(c.mods):
            # input rows padded to 16 B (zeros): the tokeniser's weight-gradient GEMM streams them by LDS-DMA (csrc/gemm_dw.hip)
            buf(f"in/{m}", (BT, _align(n, 8)), zero=True); buf(f"tgt/{m}", (BT, n), f32); buf(f"mask/{m}", (B, T), i64)
        ts, attn = buf("ts", (B, T), i64), buf("attn", (B, T), i64)
        tokmask, keypad = buf("tokmask", (B, Lq), u8), buf("keypad", (B, Lq), u8)
        keep0, mod_id, count = buf("keep0", (Lq,), u8), buf("mod_id", (Lq,), u8), buf("count", (M,), i64)
        loss_sum = buf("loss_sum", (M,), f32)

        # ---- workspaces
        max_slab = 1
        for _, n in c.mods:
            # the same arguments the launches below pass (the token embedding reads its input rows padded to 16 B: ldn selects the
            # streaming kernel and with it another split count)
            for (mm, nn, ldn) in ((n * c.mult, n, _align(n, 8)), (H, n * c.mult, None), (n, H, None)):
                S, _ = self._dw_split(mm, nn, BT, ldn=ldn)
                max_slab = max(max_slab, S * _align(mm * nn + mm))
        for (mm, nn) in ((3 * H, H), (H, H), (2 * H, H), (I, H), (H, I)):
            S, _ = self._dw_split(mm, nn, R)
            max_slab = max(max_slab, S * _align(mm * nn + mm))
        slab = buf("ws/slab", (max_slab,), f32)
        slab2 = buf("ws/slab2", (max_slab,), f32)          # a deferred weight-gradient GEMM's slabs, paired with the next one (mmfm_gemm_pair)

        def slab_fits(t, S, stride, what):
            if S * stride > t.numel():
                raise RuntimeError(f"engine: {what}: {S} slabs x {stride} floats exceed the {t.numel()}-float slab workspace")
            return t
        # launch-bound regime (R <= 8192, the reference's batch of 16): every dW GEMM keeps its own slab region and ONE
        # mmfm_reduce_slabs_multi per backward segment sums them all (66 reductions of ~7 us each otherwise)
        batch_red = R <= 8192 and os.environ.get("MMFM_BATCH_REDUCE", "1") != "0"
        pend: list = []
        slabm_off = [0]
        if batch_red:
            # regions are handed out per backward segment and reused by the next one (close_segment resets the offset behind the segment's
            # reduction): the largest segment's parameters bound the need, not the whole model's
            s_max = max(1, min(R // 256, 15))
            seg_max = max(e - s0 for _, s0, e in self.layout.segments)
            slabm = buf("ws/slabm", (s_max * (seg_max + 128 * 64),), f32)

        def slab_region(S, stride):
            o = slabm_off[0]
            slabm_off[0] = o + S * stride
            if slabm_off[0] > slabm.numel():
                raise RuntimeError(f"engine: ws/slabm holds {slabm.numel()} floats, the plan's slab regions need {slabm_off[0]}")
            return slabm[o:o + S * stride]
        maxN = max([3 * H, I] + [n * c.mult for _, n in c.mods])
        ws_col = buf("ws/col", (max(1, L.lib().mmfm_colsum_workspace(R, maxN) // 4),), f32)
        ws_ln = buf("ws/ln", (max(1, L.lib().mmfm_layernorm_bwd_workspace(R, H) // 4),), f32)
        ws_st = buf("ws/stitch", (max(1, L.lib().mmfm_stitch_bwd_workspace(code, B, T, Lq, H, c.max_F) // 4),), f32)
        ws_loss = buf("ws/loss", (max(1, L.lib().mmfm_masked_loss_workspace(BT, 1) // 4),), f32)

        def lin(plan, X, wname, Y, Mr, N, Kd, ldx=None, **kw):
            K.gemm(X, self.W(wname + ".weight"), Y, Mr, N, Kd, lda=ldx or Kd, ldb=Kd, ldc=N, bias=self.Pf(wname + ".bias"),
                   dtype=code, plan=plan, **kw)

        used_wt: list = []

        # Two weight gradients whose operands are both at hand (MLP down / up, attention out_proj / qkv) leave in ONE launch
        # (mmfm_gemm_pair): `dlin(..., defer=True)` parks the first, the next `dlin_ln` takes it along; `flush_deferred` issues a
        # parked one alone.  Each product then makes half as many K-slabs (csrc/gemm_dw.hip).
        pair_ok = code == L.BF16 and not batch_red and os.environ.get("MMFM_DW_PAIR", "1") != "0" and os.environ.get("MMFM_GEMM_DW", "1") != "0"
        deferred: list = []

        def pair_splits(Na, Ka, Nb, Kb, Mr):
            ta, tb = L.lib().mmfm_gemm_dw_tiles(Na, Ka, Mr), L.lib().mmfm_gemm_dw_tiles(Nb, Kb, Mr)
            ia = 256 * (Na + Ka) / (Na + Ka + Nb + Kb)
            Sa = max(1, int(ia) // ta)
            while Sa > 1 and (ta * Sa) % 8:
                Sa -= 1
            Sb = max(1, (256 - ta * Sa) // tb)
            out = []
            for S in (Sa, Sb):
                kchunk = _align(-(-Mr // max(1, min(S, Mr // 512))), 64)
                out += [-(-Mr // kchunk), kchunk]
            return out

        def flush_deferred(plan):
            if deferred:
                a = deferred.pop()
                dlin(plan, a["dY"], a["X"], a["wname"], a["Mr"], a["N"], a["Kd"])

        def dlin(plan, dY, X, wname, Mr, N, Kd, dX=None, ldx=None, defer=False, **kw):
            """Backward of Y[Mr,N] = X[Mr,Kd] @ W[N,Kd]^T + b:  dW, db into G;  dX = dY @ W (optional, fused epilogue).
            ldx = row stride of X when its rows are padded."""
            S, kchunk = self._dw_split(N, Kd, Mr, ldn=ldx)
            ldx = ldx or Kd
            gw, gb = self.Gv(wname + ".weight"), self.Gv(wname + ".bias")
            # bf16: the bias gradient (column sums of dY) rides on the dW GEMM (mmfm_gemm_desc.colsum); when the bias
            # gradient sits right behind the weight gradient in the flat buffer one slab reduction finishes both
            fused = code == L.BF16
            adjacent = fused and gb.data_ptr() == gw.data_ptr() + 4 * N * Kd
            if defer and pair_ok and S > 1 and adjacent and dX is None and ldx == Kd:
                flush_deferred(plan)
                deferred.append(dict(dY=dY, X=X, wname=wname, Mr=Mr, N=N, Kd=Kd))
                return
            if S == 1:
                K.gemm(dY, X, gw, N, Kd, Mr, lda=N, ldb=ldx, ldc=Kd, a_kcontig=0, b_kcontig=0, dtype=code, c_f32=1,
                       colsum=gb if fused else None, plan=plan)
            elif adjacent:
                stride = _align(N * Kd + N)
                sl = slab_region(S, stride) if batch_red else slab_fits(slab, S, stride, wname)
                K.gemm(dY, X, sl, N, Kd, Mr, lda=N, ldb=ldx, ldc=Kd, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk,
                       slab_stride=stride, dtype=code, c_f32=1, colsum=sl.data_ptr() + 4 * N * Kd, plan=plan)
                if batch_red:
                    pend.append((gw, sl, N * Kd + N, S, stride, False))
                else:
                    K.reduce_slabs(gw, sl, N * Kd + N, S, stride, plan=plan)
            else:
                sl = slab_region(S, N * Kd) if batch_red else slab_fits(slab, S, N * Kd, wname)
                K.gemm(dY, X, sl, N, Kd, Mr, lda=N, ldb=ldx, ldc=Kd, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk,
                       slab_stride=N * Kd, dtype=code, c_f32=1, plan=plan)
                if batch_red:
                    pend.append((gw, sl, N * Kd, S, N * Kd, False))
                else:
                    K.reduce_slabs(gw, sl, N * Kd, S, N * Kd, plan=plan)
            if not fused or (S > 1 and not adjacent):
                K.colsum(dY, Mr, N, N, gb, ws_col, plan=plan)
            if dX is not None:
                wT = self._w_transposed(wname, N, Kd, Mr) if fused else None
                if wT is not None:      # reduction >= 512: the 256-tile kernel (csrc/gemm_big.hip) against the K-contiguous transpose W^T [Kd, N]
                    used_wt.append(wname)
                    K.gemm(dY, wT, dX, Mr, Kd, N, lda=N, ldb=N, ldc=Kd, b_kcontig=1, dtype=code, plan=plan, **kw)
                else:
                    K.gemm(dY, self.W(wname + ".weight"), dX, Mr, Kd, N, lda=N, ldb=Kd, ldc=Kd, b_kcontig=0, dtype=code, plan=plan, **kw)

        def ln_f(plan, X, name, Y, tag, **kw):
            K.layernorm_fwd(X, self.Pf(name + ".weight"), self.Pf(name + ".bias"), Y, buf(tag + "/mean", (R,), f32),
                            buf(tag + "/rstd", (R,), f32), R, H, plan=plan, **kw)

        def ln_b(plan, dY, X, name, tag, dres, dX, **kw):
            K.layernorm_bwd(dY, X, self.b[tag + "/mean"], self.b[tag + "/rstd"], self.Pf(name + ".weight"), dres, dX,
                            self.Gv(name + ".weight"), self.Gv(name + ".bias"), R, H, ws_ln, plan=plan, **kw)

        es = 4 if self.dtype == "fp32" else 2
        scale = 1.0 / math.sqrt(dh)

        # keep decisions of the attention-probability dropout: one bit tile set per attention site, written by the forward (generator
        # kernel in front of it), read by the backward (csrc/attention_fast.hip; 51 MB per site at B = 1024).  MMFM_ATTN_KEEPBITS=0: hash.
        use_keep = self.code == L.BF16 and dp > 0 and grad and os.environ.get("MMFM_ATTN_KEEPBITS", "1") != "0"

        def attn_desc(tag, q, ldq, kv, ldkv, koff, voff, o, flags, d_o=None, dq=None, dkv=None, lddq=0, lddkv=0, dkoff=0, dvoff=0):
            keep = buf(tag + "/keep", (K.attn_keepbits_bytes(B, heads, Lq, Lq),), u8) if use_keep else None
            return K.attn_desc(code, B, heads, Lq, Lq, dh, q.data_ptr(), kv.data_ptr() + koff * es, kv.data_ptr() + voff * es, ldq, ldkv, ldkv,
                               o.data_ptr(), H, buf(tag + "/lse", (B, heads, Lq), f32), keypad, mod_id, flags, scale,
                               drop_p=self._drop(tag + "/p", dp), drop_o=self._drop(tag + "/o", dp),
                               d_o=None if d_o is None else d_o.data_ptr(), lddo=H,
                               dq=None if dq is None else dq.data_ptr(),
                               dk=None if dkv is None else dkv.data_ptr() + dkoff * es,
                               dv=None if dkv is None else dkv.data_ptr() + dvoff * es, lddq=lddq, lddk=lddkv, lddv=lddkv, keepbits=keep)

        fm = self._fused_mask(R)
        F_QKV, F_LNL, F_MLP, F_OUT = bool(fm & 1), bool(fm & 2), bool(fm & 4), bool(fm & 8)
        prep = self._build_prep() if fm else None
        if fm:
            K.prep_weights(prep["table"], prep["n"], prep["tiles"], plan=fwd)
            gdb = buf("ws/gdb", (max(_align(mm * nn + mm) for mm, nn in ((3 * H, H), (2 * H, H), (I, H), (H, H))),), f32)
            if "ws/lng" not in self.b:
                self.b["ws/lng"] = K.ln_linear_grad_workspace(H, self.device)          # zeroed once; the kernel re-arms its tickets
            ws_lng = self.b["ws/lng"]

        def ln_lin(plan, Xin, lnname, wname, Yout, N, tag, residual=None, alias=None):
            """LayerNorm + the linear it feeds in one launch; x_hat / rstd saved for the backward when training.
            alias = tag of an earlier call on the SAME input: x_hat / rstd do not depend on the LayerNorm's affine (it is folded
            into the prepared weights), so the earlier call's saved tensors serve this site's backward too and nothing is stored."""
            pw = prep["v"][wname]
            if alias is not None and grad:
                self.b[tag + "/xh"], self.b[tag + "/rs"] = self.b[alias + "/xh"], self.b[alias + "/rs"]
                xh = rs = None
            else:
                xh = buf(tag + "/xh", (R, H)) if grad else None
                rs = buf(tag + "/rs", (R,), f32) if grad else None
            K.rowgemm(Xin, pw["Wp"], Yout, R, N, H, bias=pw["bp"], ln=True, xhat=xh, rstd=rs, residual=residual,
                      ldr=H if residual is not None else 0, stream_out=True, plan=plan)

        late_lng: list = []

        def dlin_ln(plan, dYt, tag, wname, lnname, N):
            """Gradients of a LayerNorm-fed linear and of that LayerNorm's affine from G = dY^T x_hat (mmfm_ln_linear_grad)."""
            S, kchunk = self._dw_split(N, H, R)
            xh = self.b[tag + "/xh"]
            if deferred and S > 1 and deferred[-1]["Mr"] == R:
                a = deferred.pop()
                Na, Ka = a["N"], a["Kd"]
                Sa, kca, Sb, kcb = pair_splits(Na, Ka, N, H, R)
                stra, strb = _align(Na * Ka + Na), _align(N * H + N)
                slab_fits(slab2, Sa, stra, a["wname"]); slab_fits(slab, Sb, strb, wname)
                da = K.gemm_desc(a["dY"], a["X"], slab2, Na, Ka, R, lda=Na, ldb=Ka, ldc=Ka, a_kcontig=0, b_kcontig=0, splits=Sa, kchunk=kca,
                                 slab_stride=stra, dtype=code, c_f32=1, colsum=slab2.data_ptr() + 4 * Na * Ka)
                db_ = K.gemm_desc(dYt, xh, slab, N, H, R, lda=N, ldb=H, ldc=H, a_kcontig=0, b_kcontig=0, splits=Sb, kchunk=kcb,
                                  slab_stride=strb, dtype=code, c_f32=1, colsum=slab.data_ptr() + 4 * N * H)
                K.gemm_pair(da, db_, plan=plan)
                K.reduce_slabs(self.Gv(a["wname"] + ".weight"), slab2, Na * Ka + Na, Sa, stra, plan=plan)
                K.reduce_slabs(gdb, slab, N * H + N, Sb, strb, plan=plan)
            elif S == 1:
                K.gemm(dYt, xh, gdb, N, H, R, lda=N, ldb=H, ldc=H, a_kcontig=0, b_kcontig=0, dtype=code, c_f32=1,
                       colsum=gdb.data_ptr() + 4 * N * H, plan=plan)
            elif batch_red:
                # launch-bound regime: the slabs join the segment's ONE reduction launch (own region, own reduced buffer per site) and
                # mmfm_ln_linear_grad runs behind it at the end of the segment (close_segment) - one reduction launch per site less
                stride = _align(N * H + N)
                sl = slab_region(S, stride)
                g_site = buf(f"ws/gdb/{len(late_lng)}", (gdb.numel(),), f32)
                K.gemm(dYt, xh, sl, N, H, R, lda=N, ldb=H, ldc=H, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk,
                       slab_stride=stride, dtype=code, c_f32=1, colsum=sl.data_ptr() + 4 * N * H, plan=plan)
                pend.append((g_site, sl, N * H + N, S, stride, False))
                late_lng.append((g_site, wname, lnname, N))
                return
            else:
                stride = _align(N * H + N)
                slab_fits(slab, S, stride, wname)
                K.gemm(dYt, xh, slab, N, H, R, lda=N, ldb=H, ldc=H, a_kcontig=0, b_kcontig=0, splits=S, kchunk=kchunk,
                       slab_stride=stride, dtype=code, c_f32=1, colsum=slab.data_ptr() + 4 * N * H, plan=plan)
                K.reduce_slabs(gdb, slab, N * H + N, S, stride, plan=plan)
            K.ln_linear_grad(gdb, self.Pf(wname + ".weight"), self.Pf(lnname + ".weight"), self.Pf(lnname + ".bias"), N, H,
                             self.Gv(wname + ".weight"), self.Gv(wname + ".bias"), self.Gv(lnname + ".weight"), self.Gv(lnname + ".bias"),
                             ws_lng, plan=plan)

        def dx_ln(plan, dYt, Kd, tag, wname, dres, dXout):
            """dX of a LayerNorm-fed linear with the LayerNorm backward (and the residual gradient) in its epilogue."""
            K.rowgemm(dYt, prep["v"][wname]["WpT"], dXout, R, H, Kd, ldw=Kd, residual=dres, ldr=H if dres is not None else 0,
                      ln_bwd=True, bwd_xhat=self.b[tag + "/xh"], bwd_rstd=self.b[tag + "/rs"], plan=plan)

        enc_flags = L.ATTN_DIAG                                                # mm.py:152-158
        dec_flags = (L.ATTN_CAUSAL if c.causal_mask else 0) | (L.ATTN_SEP if c.sep_mask else 0)   # mm.py:178-194

        # ============================================================ forward
        K.mask_prep(B, T, [self.b[f"mask/{m}"] for m in range(M)], [1] * M, attn, [n for _, n in c.mods], tokmask, keypad, keep0,
                    mod_id, count, plan=fwd)
        tok_tmp = buf("tok_tmp", (BT, H))
        x_enc, emb_enc, x_dec = buf("x_enc", (R, H)), buf("emb_enc", (R, H)), buf("x_dec", (R, H))
        for side, xs, es_ in (("encoder", x_enc, emb_enc), ("decoder", x_dec, None)):
            for m, (mod, n) in enumerate(c.mods):
                p = f"{side}_embeddings.{mod}.embedder"
                n2 = n * c.mult
                a = buf(f"{side}/a/{m}", (BT, n2))
                # bf16 mode: the backward takes softsign' from the activation itself (act 5), no saved pre-activation (274 MB per
                # tokeniser at B = 1024, written here and read back there); the fp32 parity path keeps the exact form
                z = None if code == L.BF16 else buf(f"{side}/z/{m}", (BT, n2))
                lin(fwd, self.b[f"in/{m}"], p + ".token_embed", a, BT, n2, n, ldx=_align(n, 8), pre_out=z, act=L.ACT_SOFTSIGN, act_scale=c.embed_scale)
                lin(fwd, a, p + ".projection", tok_tmp, BT, H, n2, drop=self._drop(f"{side}/embdrop/{m}", dpe))
                mod_row = self.Pf(f"encoder_embeddings.{mod}.embedder.mod_emb.weight")[m]
                K.stitch_fwd(tok_tmp, mod_row, self.Pf(p + ".pos_embed.weight"), ts, keep0, xs, es_, B, T, Lq, m, H, c.max_F, plan=fwd)

        def out_proj(plan, a, wname, Xres, Xout):
            if F_OUT:
                K.rowgemm(a, self.W(wname + ".weight"), Xout, R, H, H, bias=self.Pf(wname + ".bias"), residual=Xres, ldr=H, plan=plan)
            else:
                lin(plan, a, wname, Xout, R, H, H, residual=Xres, ldr=H)

        def self_block(plan, X, p, tag, flags):
            """x + attn(ln1(x))  (encoder_embeddings.py:112, decoder_embeddings.py:141)."""
            qkv, a, Xa = buf(tag + "/qkv", (R, 3 * H)), buf(tag + "/a", (R, H)), buf(tag + "/xa", (R, H))
            if F_QKV:
                ln_lin(plan, X, p + ".ln1", p + ".attn.qkv", qkv, 3 * H, tag + "/ln1")
            else:
                h = buf(tag + "/h1", (R, H))
                ln_f(plan, X, p + ".ln1", h, tag + "/ln1")
                lin(plan, h, p + ".attn.qkv", qkv, R, 3 * H, H)
            K.attn_fwd(attn_desc(tag + "/sa", qkv, 3 * H, qkv, 3 * H, H, 2 * H, a, flags), plan=plan)
            out_proj(plan, a, p + ".attn.out_proj", X, Xa)
            return Xa


        def mlp_block(plan, X, p, tag):
            """x + mlp(ln2(x))  (encoder_embeddings.py:114; mm_utils.py:50-52)."""
            Xb = buf(tag + "/xb", (R, H))
            if F_MLP:
                pu = prep["v"][p + ".mlp.up_proj"]
                d_ = K.mlp_desc(R, x=X, w_up=pu["Wp"], b_up=pu["bp"], w_down=prep["v"][p + ".mlp.down_proj"]["WpP"],
                                b_down=self.Pf(p + ".mlp.down_proj.bias"), drop=self._drop(tag + "/mlpdrop", dp), y=Xb,
                                xhat=buf(tag + "/ln2/xh", (R, H)) if grad else None,
                                rstd=buf(tag + "/ln2/rs", (R,), f32) if grad else None)
                K.mlp_fwd(d_, plan=plan)
                return Xb
            h, u, g = buf(tag + "/h2", (R, H)), buf(tag + "/u", (R, I)), buf(tag + "/g", (R, I))
            ln_f(plan, X, p + ".ln2", h, tag + "/ln2")
            lin(plan, h, p + ".mlp.up_proj", g, R, I, H, pre_out=u, act=L.ACT_GELU)
            lin(plan, g, p + ".mlp.down_proj", Xb, R, H, I, drop=self._drop(tag + "/mlpdrop", dp), residual=X, ldr=H)
            return Xb

        X = x_enc
        stream_in = {}
        for i in range(c.n_enc):
            p, tag = f"encoder.{i}", f"enc{i}"
            stream_in[tag] = X
            Xa = self_block(fwd, X, p, tag, enc_flags)
            X = mlp_block(fwd, Xa, p, tag)
        enc_last = X
        enc_out, context = buf("enc_out", (R, H)), buf("context", (R, H))
        if F_LNL:
            ln_lin(fwd, X, "encoder_norm", "decoder_proj_context", context, H, "encnorm", residual=emb_enc)
        else:
            ln_f(fwd, X, "encoder_norm", enc_out, "encnorm")
            lin(fwd, enc_out, "decoder_proj_context", context, R, H, H, residual=emb_enc, ldr=H)       # mm.py:292
        Y = x_dec
        for i in range(c.n_dec):
            p, tag = f"decoder.{i}", f"dec{i}"
            stream_in[tag] = Y
            Ya = self_block(fwd, Y, p, tag, dec_flags)
            qc, kvc, a2, Yb = buf(tag + "/qc", (R, H)), buf(tag + "/kvc", (R, 2 * H)), buf(tag + "/a2", (R, H)), buf(tag + "/yb", (R, H))
            if F_LNL:
                ln_lin(fwd, Ya, p + ".query_norm", p + ".cross_attn.query", qc, H, tag + "/qn")
                # every decoder layer normalises the same context rows: the statistics are saved by the first layer only
                ln_lin(fwd, context, p + ".context_norm", p + ".cross_attn.kv", kvc, 2 * H, tag + "/cn", alias=None if i == 0 else "dec0/cn")
            else:
                hq, hc = buf(tag + "/hq", (R, H)), buf(tag + "/hc", (R, H))
                ln_f(fwd, Ya, p + ".query_norm", hq, tag + "/qn")
                ln_f(fwd, context, p + ".context_norm", hc, tag + "/cn")
                lin(fwd, hq, p + ".cross_attn.query", qc, R, H, H)
                lin(fwd, hc, p + ".cross_attn.kv", kvc, R, 2 * H, H)
            K.attn_fwd(attn_desc(tag + "/xa", qc, H, kvc, 2 * H, 0, H, a2, enc_flags), plan=fwd)   # xa_mask = encoder mask
            out_proj(fwd, a2, p + ".cross_attn.out_proj", Ya, Yb)
            Y = mlp_block(fwd, Yb, p, tag)
        dec_last = Y
        ydec = buf("ydec", (R, H))                         # de-stitched: [M][B*T][H]
        ln_f(fwd, Y, "decoder_norm", ydec, "decnorm", ds_L=Lq, ds_T=T)
        for m, (mod, n) in enumerate(c.mods):
            pred = buf(f"pred/{m}", (BT, n))
            lin(fwd, ydec[m * BT:(m + 1) * BT], f"decoder_embeddings.{mod}.out", pred, BT, n, H)
            K.masked_loss_fwd(c.loss_kind[mod], pred, self.b[f"tgt/{m}"], tokmask[:, m * T:], Lq, T, BT, n, loss_sum[m:m + 1],
                              ws_loss, plan=fwd)
        K.loss_finalize(loss_sum, count, M, self.b["loss"], self.b["inv_n"], plan=fwd)

        if not grad:
            plan = dict(fwd=fwd, bwd=None, B=B, T=T, training=bool(training), M=M, R=R, BT=BT, runs=dict(fwd=0, bwd=0), graphs={}, b=self.b)
            self.plans[key] = plan
            return plan
        # ============================================================ backward (segments fire DDP hooks)
        bwd: List[Tuple[str, list]] = []
        cur: list = []

        def close_segment(name):
            nonlocal cur
            if pend:                          # the segment's weight-gradient slabs, all in one launch, before its DDP hook fires
                K.reduce_slabs_multi(list(pend), self.device, plan=cur)
                pend.clear()
            for g_site, wname, lnname, N in late_lng:
                K.ln_linear_grad(g_site, self.Pf(wname + ".weight"), self.Pf(lnname + ".weight"), self.Pf(lnname + ".bias"), N, H,
                                 self.Gv(wname + ".weight"), self.Gv(wname + ".bias"), self.Gv(lnname + ".weight"), self.Gv(lnname + ".bias"),
                                 ws_lng, plan=cur)
            late_lng.clear()
            slabm_off[0] = 0                  # the reduction has consumed the regions (stream order): the next segment reuses them
            bwd.append((name, cur))
            cur = []

        dY = buf("d/stream", (R, H))
        dydec = buf("d/ydec", (R, H))
        t1, t2, dh_ = buf("d/t1", (R, H)), buf("d/t2", (R, H)), buf("d/h", (R, H))
        du, dqkv, dctx = buf("d/u", (R, I)), buf("d/qkv", (R, 3 * H)), buf("d/ctx", (R, H))
        for m, (mod, n) in enumerate(c.mods):
            dpred = buf(f"d/pred/{m}", (BT, n))
            K.masked_loss_bwd(c.loss_kind[mod], self.b[f"pred/{m}"], self.b[f"tgt/{m}"], tokmask[:, m * T:], Lq, T, BT, n,
                              self.b["gout"], self.b["inv_n"], dpred, plan=cur)
            dlin(cur, dpred, ydec[m * BT:(m + 1) * BT], f"decoder_embeddings.{mod}.out", BT, n, H, dX=dydec[m * BT:(m + 1) * BT])
        ln_b(cur, dydec, dec_last, "decoder_norm", "decnorm", None, dY, ds_L=Lq, ds_T=T)
        close_segment("head")

        def mlp_back(plan, dS, p, tag, X_in):
            """dS: running gradient of the residual stream (in place).  X_in = the stream value that fed ln2."""
            if F_MLP:
                pu, pdn = prep["v"][p + ".mlp.up_proj"], prep["v"][p + ".mlp.down_proj"]
                t1b, gb, dub = buf("d/t1m", (R, H)), buf("d/g", (R, I)), buf("d/du", (R, I))
                split = os.environ.get("MMFM_MLP_BWD_SPLIT", "1") == "1"   # same-box A/B at B = 1024: 30.80 -> 30.37 ms/step
                d_ = K.mlp_desc(R, w_up=pu["Wp"], b_up=pu["bp"], drop=self._drop(tag + "/mlpdrop", dp), xhat=self.b[tag + "/ln2/xh"],
                                rstd=self.b[tag + "/ln2/rs"], dy=dS, w_down_t=pdn["WpT"], w_up_t=pu["WpTP"], t1=t1b, g=gb, du=dub,
                                dx=None if split else dS)
                K.mlp_bwd(d_, plan=plan)
                if split:     # front half only above (t1, g, du); dX + LayerNorm backward + residual by the row-owner K = I kernel
                    dx_ln(plan, dub, I, tag + "/ln2", p + ".mlp.up_proj", dS, dS)
                dlin(plan, t1b, gb, p + ".mlp.down_proj", R, H, I, defer=True)     # dW_down = t1^T g, db_down = colsum t1
                dlin_ln(plan, dub, tag + "/ln2", p + ".mlp.up_proj", p + ".ln2", I)
                flush_deferred(plan)
                return
            dSd = dS
            if dp > 0:                                                       # mm_utils.py:52 dropout(down_proj(.))
                K.dropout_apply(dS, t1, R, H, self._drop(tag + "/mlpdrop", dp), plan=plan)
                dSd = t1
            dlin(plan, dSd, self.b[tag + "/g"], p + ".mlp.down_proj", R, H, I, dX=du, act=L.ACT_GELU_GRAD, gradmul_pre=self.b[tag + "/u"])
            dlin(plan, du, self.b[tag + "/h2"], p + ".mlp.up_proj", R, I, H, dX=dh_)
            ln_b(plan, dh_, X_in, p + ".ln2", tag + "/ln2", dS, dS)

        def out_proj_back(plan, dS, a, wname):
            """dW, db of an attention out_proj and d(attention output) -> t2."""
            if F_OUT:
                dlin(plan, dS, a, wname, R, H, H, defer=True)        # leaves with the next LayerNorm-fed linear's weight gradient
                K.rowgemm(dS, prep["v"][wname]["WpT"], t2, R, H, H, plan=plan)
            else:
                dlin(plan, dS, a, wname, R, H, H, dX=t2)

        def self_back(plan, dS, p, tag, X_in, flags):
            out_proj_back(plan, dS, self.b[tag + "/a"], p + ".attn.out_proj")
            qkv = self.b[tag + "/qkv"]
            K.attn_bwd(attn_desc(tag + "/sa", qkv, 3 * H, qkv, 3 * H, H, 2 * H, self.b[tag + "/a"], flags, d_o=t2, dq=dqkv, dkv=dqkv,
                                 lddq=3 * H, lddkv=3 * H, dkoff=H, dvoff=2 * H), plan=plan)
            if F_QKV:
                dlin_ln(plan, dqkv, tag + "/ln1", p + ".attn.qkv", p + ".ln1", 3 * H)
                flush_deferred(plan)
                dx_ln(plan, dqkv, 3 * H, tag + "/ln1", p + ".attn.qkv", dS, dS)
            else:
                flush_deferred(plan)
                dlin(plan, dqkv, self.b[tag + "/h1"], p + ".attn.qkv", R, 3 * H, H, dX=dh_)
                ln_b(plan, dh_, X_in, p + ".ln1", tag + "/ln1", dS, dS)

        dqc, dkvc = buf("d/qc", (R, H)), buf("d/kvc", (R, 2 * H))
        first_ctx = True
        for i in reversed(range(c.n_dec)):
            p, tag = f"decoder.{i}", f"dec{i}"
            mlp_back(cur, dY, p, tag, self.b[tag + "/yb"])
            # cross attention (decoder_embeddings.py:143): query side -> stream, context side -> dctx
            out_proj_back(cur, dY, self.b[tag + "/a2"], p + ".cross_attn.out_proj")
            K.attn_bwd(attn_desc(tag + "/xa", self.b[tag + "/qc"], H, self.b[tag + "/kvc"], 2 * H, 0, H, self.b[tag + "/a2"], enc_flags,
                                 d_o=t2, dq=dqc, dkv=dkvc, lddq=H, lddkv=2 * H, dkoff=0, dvoff=H), plan=cur)
            if F_LNL:
                dlin_ln(cur, dqc, tag + "/qn", p + ".cross_attn.query", p + ".query_norm", H)
                flush_deferred(cur)
                dx_ln(cur, dqc, H, tag + "/qn", p + ".cross_attn.query", dY, dY)
                dlin_ln(cur, dkvc, tag + "/cn", p + ".cross_attn.kv", p + ".context_norm", 2 * H)
                dx_ln(cur, dkvc, 2 * H, tag + "/cn", p + ".cross_attn.kv", None if first_ctx else dctx, dctx)
            else:
                flush_deferred(cur)
                dlin(cur, dqc, self.b[tag + "/hq"], p + ".cross_attn.query", R, H, H, dX=dh_)
                ln_b(cur, dh_, self.b[tag + "/xa"], p + ".query_norm", tag + "/qn", dY, dY)
                dlin(cur, dkvc, self.b[tag + "/hc"], p + ".cross_attn.kv", R, 2 * H, H, dX=dh_)
                ln_b(cur, dh_, context, p + ".context_norm", tag + "/cn", None if first_ctx else dctx, dctx)
            first_ctx = False
            self_back(cur, dY, p, tag, stream_in[tag], dec_flags)
            close_segment(p)
        if c.n_dec == 0:
            raise NotImplementedError("n_dec == 0")
        # now dY = d(dec_tokens + dec_emb) and dctx = d(context); context = ctx_proj(enc_out) + encoder_emb (mm.py:292)
        dX = buf("d/xstream", (R, H))
        if F_LNL:
            dlin_ln(cur, dctx, "encnorm", "decoder_proj_context", "encoder_norm", H)
            dx_ln(cur, dctx, H, "encnorm", "decoder_proj_context", None, dX)
        else:
            dlin(cur, dctx, enc_out, "decoder_proj_context", R, H, H, dX=dh_)
            ln_b(cur, dh_, enc_last, "encoder_norm", "encnorm", None, dX)
        close_segment("bridge")
        for i in reversed(range(c.n_enc)):
            p, tag = f"encoder.{i}", f"enc{i}"
            mlp_back(cur, dX, p, tag, self.b[tag + "/xa"])
            self_back(cur, dX, p, tag, stream_in[tag], enc_flags)
            close_segment(p)
        # tokenisers: decoder side first (it overwrites the shared mod_emb gradient row, the encoder side adds)
        for side, dS, dextra, acc_mod in (("decoder", dY, None, False), ("encoder", dX, dctx, True)):
            for m, (mod, n) in enumerate(c.mods):
                pS = f"{side}_embeddings.{mod}.embedder"
                K.stitch_bwd(dS, dextra, ts, keep0, self._drop(f"{side}/embdrop/{m}", dpe), buf(f"d/tok/{side}/{m}", (BT, H)),
                             self.Gv(f"encoder_embeddings.{mod}.embedder.mod_emb.weight")[m], self.Gv(pS + ".pos_embed.weight"),
                             acc_mod, False, B, T, Lq, m, H, c.max_F, ws_st, plan=cur)
        for side in ("decoder", "encoder"):
            for m, (mod, n) in enumerate(c.mods):
                p = f"{side}_embeddings.{mod}.embedder"
                n2 = n * c.mult
                dz = buf(f"d/z/{m}", (BT, n2))
                if code == L.BF16:
                    dlin(cur, self.b[f"d/tok/{side}/{m}"], self.b[f"{side}/a/{m}"], p + ".projection", BT, H, n2, dX=dz,
                         act=L.ACT_SOFTSIGN_GRAD_OUT, act_scale=c.embed_scale, gradmul_pre=self.b[f"{side}/a/{m}"])
                else:
                    dlin(cur, self.b[f"d/tok/{side}/{m}"], self.b[f"{side}/a/{m}"], p + ".projection", BT, H, n2, dX=dz,
                         act=L.ACT_SOFTSIGN_GRAD, act_scale=c.embed_scale, gradmul_pre=self.b[f"{side}/z/{m}"])
                dlin(cur, dz, self.b[f"in/{m}"], p + ".token_embed", BT, n2, n, ldx=_align(n, 8))
        close_segment("embed")
        if used_wt:                 # refresh the bf16 transposes once per step, in front of everything (the optimiser rewrote the weights)
            tw = self._wt_table()
            K.prep_weights(tw["table"], tw["n"], tw["tiles"], plan=fwd)
            fwd.insert(0, fwd.pop())
        plan = dict(fwd=fwd, bwd=bwd, B=B, T=T, training=bool(training), M=M, R=R, BT=BT, runs=dict(fwd=0, bwd=0), graphs={}, b=self.b)
        self.plans[key] = plan
        return plan

    def _run(self, plan, which, entries_fn, tag=None):
        """Run a piece of the plan: eagerly the first time (lazy one-off initialisation such as the >64 KB LDS
        opt-in must not happen inside a capture), then capture it into a hipGraph and replay."""
        tag = which if tag is None else tag
        g = plan["graphs"].get(tag)
        if g is not None:
            g.replay()
            return
        if not self.use_graphs or plan["runs"][which] < 1:
            entries_fn()
            return
        graph = torch.cuda.CUDAGraph()
        # thread_local: with the DDP wrapper the previous bucket's all-reduce may still be running on RCCL's stream while the next
        # backward segment is captured; the default (global) mode would treat that foreign-stream activity as a capture violation
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            entries_fn()
        plan["graphs"][tag] = graph
        graph.replay()

    # ------------------------------------------------------------------ data in
    def load_inputs(self, B, T, inputs, targets, masks, ts, attn):
        """Copy one batch into the static input buffers (device tensors or host tensors)."""
        for m in range(len(self.cfg.mods)):
            self.b[f"in/{m}"].view(B, T, -1)[..., :inputs[m].shape[-1]].copy_(inputs[m], non_blocking=True)
            self.b[f"tgt/{m}"].view(B, T, -1).copy_(targets[m], non_blocking=True)
            self.b[f"mask/{m}"].copy_(masks[m], non_blocking=True)
        self.b["ts"].copy_(ts, non_blocking=True)
        self.b["attn"].copy_(attn, non_blocking=True)

    # ------------------------------------------------------------------ run
    def forward(self, B, T, inputs, targets, masks, ts, attn, training=True, anchor=None):
        want_grad = anchor is not None and torch.is_grad_enabled() and anchor.requires_grad
        plan = self._plan(B, T, training, want_grad)
        self.load_inputs(B, T, inputs, targets, masks, ts, attn)
        advance = training and (self.cfg.dropout > 0 or self.cfg.embed_dropout > 0)

        def fwd_entries():
            if advance:
                K.rng_advance(self.rng)
            K.run_plan(plan["fwd"])
        self._run(plan, "fwd", fwd_entries)
        plan["runs"]["fwd"] += 1
        self._token += 1
        self._fwd_token = self._token
        self._last = plan
        M = plan["M"]
        out = dict(mod_loss=[self.b["loss_sum"][m].clone() for m in range(M)],
                   mod_n=[self.b["count"][m].clone() for m in range(M)],
                   preds=[self.b[f"pred/{m}"].view(B, T, -1) for m in range(M)])
        if want_grad:
            out["loss"] = _StepFn.apply(anchor, self, self._token)
        else:
            out["loss"] = self.b["loss"].clone().reshape(())
        return out

    def backward(self, grad_out=None, token=None):
        if token is not None and token != self._fwd_token:
            raise RuntimeError("backward() must follow the forward() that produced this loss: the engine keeps one "
                               "set of activation buffers")
        if grad_out is None:
            self.b["gout"].fill_(1.0)
        else:
            self.b["gout"].copy_(grad_out.reshape(1).to(torch.float32))
        accumulate_into = None
        first = next(iter(self.params.values()), None)
        if first is not None and first.grad is not None:
            accumulate_into = self.G.clone()               # caller did not zero_grad(): keep torch's += semantics
        plan = self._last
        if plan["bwd"] is None:
            raise RuntimeError("backward(): the last forward ran without gradient tracking (forward-only plan)")
        if (plan["B"], plan["T"]) not in self._pools or self._pools[(plan["B"], plan["T"])] is not plan["b"]:
            raise RuntimeError("backward(): the batch shape of this loss was evicted (MMFM_MAX_SHAPES) before its backward ran")
        self.b = plan["b"]
        if self.grad_ready_hooks and accumulate_into is None:
            for name, seg in plan["bwd"]:          # DDP: one graph per segment, collectives issued in between
                self._run(plan, "bwd", lambda seg=seg: K.run_plan(seg), tag="bwd/" + name)
                for hook in self.grad_ready_hooks:
                    hook(name)
        else:
            self._run(plan, "bwd", lambda: [K.run_plan(seg) for _, seg in plan["bwd"]])
        plan["runs"]["bwd"] += 1
        if accumulate_into is not None:
            self.G.add_(accumulate_into)
            for hook in self.grad_ready_hooks:
                for name, _ in self._last["bwd"]:
                    hook(name)
        for hook in self.backward_done_hooks:
            hook()
        for name, p in self.params.items():
            if p.grad is None or p.grad.data_ptr() != self.Gv(name).data_ptr():
                p.grad = self.Gv(name)

    def segment_range(self, name):
        for n, s, e in self.layout.segments:
            if n == name:
                return s, e
        raise KeyError(name)

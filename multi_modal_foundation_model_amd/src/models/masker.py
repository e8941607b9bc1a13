"""Random mask generator (reference: models/masker.py:56-174).

Host-side by design: the reference draws from torch's CPU generator (and Python's `random` for
region sampling), and the masked-pretraining path keeps only `mask[:, :, 0]` (mm.py:267-270), so
moving this to the GPU would change the random stream for no gain.  The order and shapes of the
generator calls below are what makes the masks bit-exact against the reference
(tests/golden/masker_bits.npz): bernoulli(expand_prob) [, randint], bernoulli(mask_probs),
bernoulli(zero_ratio x shape), bernoulli(random_ratio x shape), rand(shape) — all on the CPU
generator, wherever `spikes` lives.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

TOKEN_MODES = ("temporal", "random_token", "causal")


class _PinnedRing:
    """Asynchronous host->device copies of small tensors.  `t.to(device)` from pageable memory is a synchronous
    copy that drains the stream (a ~5 ms stall per call once the GPU runs ahead); staging through pinned
    buffers makes it truly asynchronous.  A slot is reused only after the copy that last read it completed."""

    def __init__(self, slots=8):
        self.slots, self.i, self.bufs, self.events = slots, 0, {}, {}

    def to(self, t, device):
        device = torch.device(device)
        if device.type != "cuda" or t.is_cuda:
            return t.to(device)
        key = (tuple(t.shape), t.dtype)
        if key not in self.bufs:
            self.bufs[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(self.slots)]
            self.events[key] = [None] * self.slots
        self.i = (self.i + 1) % self.slots
        ev = self.events[key][self.i]
        if ev is not None:
            ev.synchronize()
        buf = self.bufs[key][self.i]
        buf.copy_(t)
        out = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[key][self.i] = ev
        return out


class Masker(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.force_active = config.force_active if "force_active" in config else False
        self.mode = config.mode
        self.ratio = config.ratio
        self.zero_ratio = config.zero_ratio
        self.random_ratio = config.random_ratio
        self.expand_prob = config.expand_prob
        self.max_timespan = config.max_timespan
        self.channels = config.channels
        self.timesteps = config.timesteps
        self.mask_regions = config.mask_regions
        self.target_regions = config.target_regions
        self.n_mask_regions = config.n_mask_regions
        self.causal_zero = config.causal_zero
        # Throughput switch of ONE training run, set by the trainer (trainer/base.py: it IS the trainer's default for
        # mask_type 'embd'; `training.exact_masker_stream: true` / MMFM_EXACT_MASKER=1 keeps it off): in the `embd` masking
        # path the caller discards the corrupted spikes, so only the token-level draw matters.  Skipping the three full-size
        # draws shortens the walk through the CPU generator: masks stay identically distributed but, from the second masker
        # call on, are NOT the reference's bit for bit.  It is honoured only by callers that discard the spikes (they pass it
        # per call, mm.py) and it is never pickled: a checkpoint always starts on the reference's stream.
        self.token_mask_only = False
        self._ring = _PinnedRing()

    def __getstate__(self):                    # checkpoints pickle the whole model: drop pinned buffers / events
        state = self.__dict__.copy()
        state["_ring"] = None
        state["token_mask_only"] = False       # a run's throughput switch must not leak into model_best.pt / model_last.pt
        return state

    @staticmethod
    def _no_mask(spikes):
        return spikes, torch.zeros_like(spikes).to(torch.int64)

    def forward(self, spikes, neuron_regions=None, token_mask_only=False):
        """`token_mask_only=True` (callers that discard the returned spikes only): skip the zero / random corruption draws."""
        inactive = (not self.training and not self.force_active) or self.target_regions is None \
            or self.mask_regions is None or self.ratio == 0
        if inactive:
            return self._no_mask(spikes)
        if "all" in self.mask_regions:
            self.mask_regions = list(np.unique(neuron_regions))
        if "all" in self.target_regions:
            self.target_regions = list(np.unique(neuron_regions))

        B, T, N = spikes.shape
        dev = spikes.device
        ratio = self.ratio
        targets_sel = None
        if self.mode in TOKEN_MODES:
            timespan = 1
            if torch.bernoulli(torch.tensor(self.expand_prob).float()):
                timespan = torch.randint(1, self.max_timespan + 1, (1,)).item()
            probs = torch.full((B, T), ratio / timespan)
            if self.mode == "causal":
                timespan = torch.randint(1, self.max_timespan + 1, (1,)).item()
                probs = torch.full((B, T), 0.01)
        elif self.mode == "neuron":
            probs = torch.full((B, N), ratio)
        elif self.mode == "random":
            probs = torch.full((B, T, N), ratio)
        elif self.mode == "co-smooth":
            assert self.channels is not None, "No channels to mask"
            probs = torch.zeros(N)
            probs[list(self.channels)] = 1
        elif self.mode == "forward-pred":
            assert self.timesteps is not None, "No time steps to mask"
            probs = torch.zeros(T)
            probs[list(self.timesteps)] = 1
        elif self.mode == "inter-region":
            assert neuron_regions is not None, "Can't mask region without brain region information"
            probs = torch.zeros(B, N)
            for region in random.sample(self.mask_regions, self.n_mask_regions):
                probs[torch.tensor(neuron_regions == region)] = 1
        elif self.mode == "intra-region":
            assert neuron_regions is not None, "Can't mask region without brain region information"
            probs = torch.ones(B, N)
            targets_sel = torch.zeros(B, N)
            for region in random.sample(self.target_regions, self.n_mask_regions):
                sel = torch.tensor(neuron_regions == region)
                probs[sel] = ratio
                targets_sel[sel] = 1
        else:
            raise Exception(f"Masking mode {self.mode} not implemented")

        if self._ring is None:
            self._ring = _PinnedRing()
        drawn = self._ring.to(torch.bernoulli(probs), dev)
        causal_target = None
        if self.mode in TOKEN_MODES:
            if timespan > 1:
                drawn = self.expand_timesteps(drawn, timespan)
            if self.causal_zero and self.mode == "causal":
                first = torch.argmax(drawn.int(), dim=1).int()
                causal_target = drawn.clone()
                for b in range(B):
                    drawn[b, first[b]:] = 1
            mask = drawn.unsqueeze(2).expand(B, T, N).bool()
        elif self.mode in ("neuron", "region", "intra-region", "inter-region"):
            mask = drawn.unsqueeze(1).expand(B, T, N).bool()
        elif self.mode == "co-smooth":
            mask = drawn[None, None, :].expand(B, T, N).bool()
        elif self.mode == "forward-pred":
            mask = drawn[None, :, None].expand(B, T, N).bool()
        else:
            mask = drawn.bool()

        if not token_mask_only:
            zero_idx = torch.bernoulli(torch.full((B, T, N), float(self.zero_ratio))).to(dev).bool() & mask
            spikes[zero_idx] = 0
            rand_idx = torch.bernoulli(torch.full((B, T, N), float(self.random_ratio))).to(dev).bool() & mask & ~zero_idx
            # CPU generator on purpose (the reference draws this one on `spikes.device`; the CPU oracle that
            # produced the fixtures has its data on the CPU, SURVEY.md §7 "Masker RNG")
            noise = (spikes.max() * torch.rand((B, T, N)).to(dev)).to(spikes.dtype)
            spikes[rand_idx] = noise[rand_idx]

        if causal_target is not None:
            targets_mask = causal_target.unsqueeze(2).expand(B, T, N).bool()
        elif self.mode == "intra-region":
            targets_mask = mask & targets_sel.unsqueeze(1).expand(B, T, N).bool().to(dev)
        else:
            targets_mask = mask
        return spikes, targets_mask.to(torch.int64)

    @staticmethod
    def expand_timesteps(mask, width=1):
        kernel = torch.ones(width, device=mask.device).view(1, 1, -1)
        return F.conv1d(mask.unsqueeze(1), kernel, padding="same").squeeze(1) >= 1

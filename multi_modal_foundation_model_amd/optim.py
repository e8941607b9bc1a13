"""Fused AdamW over the engine's flat parameter buffer (one HIP launch per step).

Drop-in for `torch.optim.AdamW(model.parameters(), lr, weight_decay, eps)`
(reference: train_multi_modal.py:197-202, stepping at trainer/base.py:196-198): same constructor
arguments, same `param_groups` (so `OneCycleLR` keeps rewriting `lr` and `betas[0]` every step),
same update rule as torch's single-tensor AdamW.  `zero_grad()` is O(1): the next backward
overwrites the flat gradient buffer.
"""
from __future__ import annotations

import math

import torch

from . import ops as K


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, engine=None, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.engine = engine
        self.grad_scale = grad_scale
        self._t = 0
        self._m = self._v = self._hyper = None
        self._ring, self._ring_ev, self._ring_i = [], [], 0      # pinned staging slots for the per-step scalars
        self.pre_step_hooks = []        # e.g. the DDP wrapper's "wait for the all-reduce" hook

    def attach(self, engine):
        self.engine = engine
        return self

    def _bind(self):
        eng = self.engine
        if eng is None:
            raise RuntimeError("FusedAdamW needs the model's engine: call opt.attach(model.engine()) after the first forward, "
                               "or construct it through make_optimizer(model, ...)")
        if self._m is None or self._m.numel() != eng.P.numel() or self._m.device != eng.P.device:
            self._m, self._v = torch.zeros_like(eng.P), torch.zeros_like(eng.P)
            self._hyper = torch.zeros(8, device=eng.P.device)
            # the host may run many steps ahead of the GPU: each async H2D copy of the scalars gets its own
            # pinned slot, reused only after the copy that last read it has completed
            self._ring = [torch.zeros(8).pin_memory() for _ in range(16)]
            self._ring_ev = [None] * 16
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW handles one param group (the reference builds exactly one)")
        mine = {id(p) for p in self.param_groups[0]["params"]}
        theirs = {id(p) for p in eng.params.values()}
        if mine != theirs:
            raise RuntimeError("FusedAdamW must own exactly the engine's parameters (model.parameters())")
        return eng

    # ------------------------------------------------------------------ checkpoint (resume, absent upstream)
    def state_dict(self):
        """torch's state_dict plus the fused state: step count and the flat first / second moment buffers (engine layout).
        Without them a resume would restart the bias correction at t = 1 with zero moments."""
        sd = super().state_dict()
        sd["fused"] = dict(t=self._t, m=None if self._m is None else self._m.detach().cpu().clone(),
                           v=None if self._v is None else self._v.detach().cpu().clone())
        return sd

    def load_state_dict(self, state_dict):
        fused = state_dict.get("fused")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "fused"})
        if fused is None:
            raise KeyError("FusedAdamW.load_state_dict: no 'fused' entry (moments / step count): not a FusedAdamW checkpoint")
        self._t = int(fused["t"])
        if fused["m"] is not None:
            eng = self._bind()
            if fused["m"].numel() != eng.P.numel():
                raise ValueError(f"checkpoint moments have {fused['m'].numel()} elements, the engine's flat buffer {eng.P.numel()}")
            self._m.copy_(fused["m"].to(self._m.device))
            self._v.copy_(fused["v"].to(self._v.device))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._bind()
        # torch skips parameters whose .grad is None: after zero_grad() and before the next backward that is all of them
        # (the flat gradient buffer still holds the previous step's values - they must not be applied twice)
        if all(p.grad is None for p in self.param_groups[0]["params"]):
            return loss
        for hook in self.pre_step_hooks:
            hook()
        g = self.param_groups[0]
        self._t += 1
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        bc1, bc2 = 1.0 - b1 ** self._t, 1.0 - b2 ** self._t
        i = self._ring_i = (self._ring_i + 1) % len(self._ring)
        if self._ring_ev[i] is not None:
            self._ring_ev[i].synchronize()
        h = self._ring[i]
        h[0], h[1], h[2], h[3] = 1.0 - lr * wd, 1.0 - b1, b2, 1.0 - b2
        h[4], h[5], h[6], h[7] = lr / bc1, math.sqrt(bc2), eps, self.grad_scale
        self._hyper.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[i] = ev
        K.adamw_step(eng.P, eng.G, self._m, self._v, eng.Pw if eng.dtype == "bf16" else None, eng.P.numel(), self._hyper)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        for p in self.param_groups[0]["params"]:
            p.grad = None


def make_optimizer(model, lr, weight_decay, eps):
    """What train_multi_modal.py:197-202 builds, fused.  The engine is bound lazily at the first step."""
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay, eps=eps)
    inner = getattr(model, "module", model)
    opt._model = inner
    orig_bind = opt._bind

    def bind():
        if opt.engine is None or opt.engine is not inner._engine:
            opt.engine = inner.engine()
        return orig_bind()
    opt._bind = bind
    return opt

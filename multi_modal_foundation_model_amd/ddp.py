"""Data parallelism for the engine: bucketed gradient all-reduce on RCCL over xGMI, overlapped
with the rest of backward.

Replaces what `accelerator.prepare(model)` becomes under a multi-process launch in the reference
(torch DDP, train_multi_modal.py:177,195).  Semantics kept: every rank runs its own batch through
a full replica and the gradient applied is the MEAN over ranks of the per-rank gradients of the
per-rank-normalised losses (mm.py:237) — not the gradient of a globally normalised loss.  One
process per GPU; the only collective on the path is this all-reduce (9.4 M fp32 values = 37.6 MB).

MI355X specifics: the flat gradient buffer is laid out in forward order, so backward completes it
back to front and each bucket is ONE contiguous range -> one large collective per bucket, no
gather/scatter copies.  xGMI is point-to-point (7 links/GPU): a few multi-MB buckets keep every
link busy while backward continues; the last bucket (the tokenisers) cannot overlap.  RCCL runs
the collective on its own stream; `work.wait()` only makes the compute stream wait, not the host.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def backward_order(layout, cfg) -> List[str]:
    return (["head"] + [f"decoder.{i}" for i in reversed(range(cfg.n_dec))] + ["bridge"]
            + [f"encoder.{i}" for i in reversed(range(cfg.n_enc))] + ["embed"])


class GradBuckets:
    """Contiguous ranges of the flat gradient buffer, in the order backward completes them."""

    def __init__(self, layout, cfg, bucket_bytes: int = 8 << 20):
        seg = {n: (s, e) for n, s, e in layout.segments}
        order = backward_order(layout, cfg)
        self.buckets = []          # (last_segment_name, start, end)
        cur_hi = cur_lo = None
        for name in order:
            s, e = seg[name]
            if cur_hi is None:
                cur_lo, cur_hi = s, e
            else:
                assert e == cur_lo or e <= cur_lo, "segments must be adjacent, descending"
                cur_lo = s
            if (cur_hi - cur_lo) * 4 >= bucket_bytes or name == order[-1]:
                self.buckets.append((name, cur_lo, cur_hi))
                cur_hi = cur_lo = None
        self.trigger = {name: (lo, hi) for name, lo, hi in self.buckets}


class EngineDDP:
    """Attach to an Engine: all-reduce each gradient bucket as soon as backward has produced it."""

    def __init__(self, engine, process_group=None, bucket_bytes: int = 8 << 20, broadcast: bool = True):
        self.engine, self.pg = engine, process_group
        self.world = dist.get_world_size(process_group)
        self.buckets = GradBuckets(engine.layout, engine.cfg, bucket_bytes)
        self.works = []
        backend = dist.get_backend(process_group)
        self.avg_op = dist.ReduceOp.AVG if backend == "nccl" else None
        if broadcast:                       # replicas start identical (rank 0's parameters), like torch DDP
            dist.broadcast(engine.P, src=0, group=process_group)
            engine.refresh_weights()
        engine.grad_ready_hooks.append(self.on_segment)
        engine.backward_done_hooks.append(self.finish)

    def on_segment(self, name: str):
        rng = self.buckets.trigger.get(name)
        if rng is None:
            return
        lo, hi = rng
        g = self.engine.G[lo:hi]
        if self.avg_op is not None:
            self.works.append((dist.all_reduce(g, op=self.avg_op, group=self.pg, async_op=True), None))
        else:
            self.works.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), g))

    def finish(self):
        for w, g in self.works:
            w.wait()
            if g is not None:
                g.mul_(1.0 / self.world)
        self.works = []


class DataParallelModel(torch.nn.Module):
    """What `Accelerator.prepare(model)` returns under a multi-process launch."""

    def __init__(self, module, process_group=None, bucket_bytes: int = 8 << 20):
        super().__init__()
        self.module = module
        self._pg, self._bucket_bytes, self._ddp = process_group, bucket_bytes, None

    def forward(self, *a, **kw):
        eng = self.module.engine()           # builds / re-adopts the engine from the module's current parameters
        if self._ddp is None or self._ddp.engine is not eng:
            # like torch DDP's constructor: rank 0's parameters are broadcast BEFORE the first forward, so every replica's
            # first loss and gradient are taken at the same parameters even if the replicas were initialised differently
            self._ddp = EngineDDP(eng, self._pg, self._bucket_bytes, broadcast=True)
        return self.module(*a, **kw)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


class Accelerator:
    """The slice of accelerate.Accelerator the reference uses: `.device` and `.prepare(model)`
    (train_multi_modal.py:177,195; trainer/base.py:55,60-61,257).  One process per GPU; rank and
    world size come from the torchrun environment."""

    def __init__(self, backend: Optional[str] = None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
            self.device = torch.device("cpu")
        if self.world > 1 and not dist.is_initialized():
            dist.init_process_group(backend or ("nccl" if self.device.type == "cuda" else "gloo"))

    @property
    def is_main_process(self):
        return self.rank == 0

    def prepare(self, model):
        model = model.to(self.device)
        return DataParallelModel(model) if self.world > 1 else model

"""Evaluation metrics on the device: drop-ins for the host loops of the reference's eval path.

* `r2_series(gt, pred)`      - what `utils/utils.py:107-115` computes with 50 x B torcheval calls: R^2 of every (g, c) series
                               of a [G, S, C] pair over S, one launch, strided views accepted as they are.
* `trial_avg_r2(gt, pred)`   - `metrics_list(..., metrics=["r2"])["r2"]`: nan/inf-masked mean over c, mean over g.
* `bits_per_spike(rates, spikes)` - `utils/eval_utils.py:1095-1119` (NLB co-smoothing metric).

* `bits_per_spike_per_neuron(rates, spikes)` - the per-neuron loop of `spiking_activity_recon_eval`
                               (`utils/eval_utils.py:846-851`, N host calls upstream) in one pass; inf -> nan like upstream.

There is no CPU path here: tensors must live on the GPU (the CPU restatement is oracle/metrics_oracle.py, test-only).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .ops import P, _stream as stream


def _need_cuda(*ts):
    for t in ts:
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise RuntimeError("multi_modal_foundation_model_amd.metrics runs on the MI355X; move the tensors to the GPU")


def r2_series(gt: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    """gt, pred: [G, S, C] fp32 CUDA tensors (any strides).  Returns [G, C] fp32: 1 - SS_res / SS_tot over S."""
    _need_cuda(gt, pred)
    if gt.shape != pred.shape or gt.dim() != 3:
        raise ValueError(f"r2_series: expected two [G, S, C] tensors, got {tuple(gt.shape)} and {tuple(pred.shape)}")
    gt = gt if gt.dtype == torch.float32 else gt.float()
    pred = pred if pred.dtype == torch.float32 else pred.float()
    G, S, Cn = gt.shape
    out = torch.empty(G, Cn, device=gt.device, dtype=torch.float32)
    gs, ps = (C.c_int64 * 3)(*gt.stride()), (C.c_int64 * 3)(*pred.stride())
    L.check(L.lib().mmfm_r2_series(P(gt), gs, P(pred), ps, G, S, Cn, P(out), stream()), "mmfm_r2_series")
    return out


def trial_avg_r2(gt: torch.Tensor, pred: torch.Tensor) -> float:
    """metrics_list(gt, pred, metrics=["r2"])["r2"] (utils/utils.py:109-115): for every g the invalid-masked mean over the
    c series of R^2 over S, then the mean over g."""
    vals = r2_series(gt, pred).double().cpu().numpy()
    return float(np.mean([np.ma.masked_invalid(v).mean() for v in vals]))


def bits_per_spike(rates, spikes) -> float:
    """utils/eval_utils.py:1095-1119 on the device.  rates, spikes: [..., N] (trials x bins x neurons), same shape."""
    rates, spikes = torch.as_tensor(rates), torch.as_tensor(spikes)
    _need_cuda(rates, spikes)
    if rates.shape != spikes.shape:
        raise AssertionError(f"neg_log_likelihood: Rates and spikes should be of the same shape. spikes: {tuple(spikes.shape)}, "
                             f"rates: {tuple(rates.shape)}")
    N = rates.shape[-1]
    r = rates.reshape(-1, N).float().contiguous()
    s = spikes.reshape(-1, N).float().contiguous()
    R = r.shape[0]
    ws = torch.empty(L.lib().mmfm_bits_per_spike_workspace(R, N), dtype=torch.uint8, device=r.device)
    out = torch.empty(4, device=r.device, dtype=torch.float32)
    L.check(L.lib().mmfm_bits_per_spike(P(r), P(s), R, N, P(out), P(ws), ws.numel(), stream()), "mmfm_bits_per_spike")
    return float(out[0].item())


def bits_per_spike_per_neuron(rates, spikes) -> torch.Tensor:
    """[..., N] rates / spikes -> [N] fp32 (on the device): bits_per_spike(rates[..., [n]], spikes[..., [n]]) for every n,
    with upstream's `if np.isinf(bps): bps = np.nan` applied (utils/eval_utils.py:846-851)."""
    rates, spikes = torch.as_tensor(rates), torch.as_tensor(spikes)
    _need_cuda(rates, spikes)
    if rates.shape != spikes.shape:
        raise AssertionError("neg_log_likelihood: Rates and spikes should be of the same shape.")
    N = rates.shape[-1]
    r = rates.reshape(-1, N).float().contiguous()
    s = spikes.reshape(-1, N).float().contiguous()
    R = r.shape[0]
    ws = torch.empty(L.lib().mmfm_bits_per_spike_neurons_workspace(R, N), dtype=torch.uint8, device=r.device)
    out = torch.empty(N, device=r.device, dtype=torch.float32)
    L.check(L.lib().mmfm_bits_per_spike_neurons(P(r), P(s), R, N, P(out), P(ws), ws.numel(), stream()), "mmfm_bits_per_spike_neurons")
    return torch.where(torch.isinf(out), torch.full_like(out, float("nan")), out)

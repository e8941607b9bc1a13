"""The slice of the reference's `utils/eval_utils.py` that sits next to the hot path: the held-out
mask construction of the evaluation modes and the spike-prediction metrics.  Plotting, PSTH analysis and the dataset-bound
drivers (`co_smoothing_eval`, `load_model_data_local`, ...) are out of scope.

* `heldout_mask`          - utils/eval_utils.py:988-1045: which (trial, bin, neuron) entries a co-smoothing /
                            forward-prediction / inter- / intra-region evaluation hides; index ops only, runs wherever
                            `spike_data` lives.  Bit-exact against the reference's function (tests/test_metrics.py).
* `bits_per_spike`, `bits_per_spike_per_neuron`, `neg_log_likelihood` - utils/eval_utils.py:1051-1119 and the per-neuron
                            loop of `spiking_activity_recon_eval` (:846-851), computed by the HIP kernels in csrc/metrics.hip.
"""
import numpy as np
import torch


def heldout_mask(spike_data, mode='manual', heldout_idxs=np.array([]), n_active=1, target_regions=None, neuron_regions=None):
    """Returns {"spikes": spike_data with the held-out entries zeroed, "heldout_idxs": the held-out indices,
    "eval_mask": 1 where an entry is held out}.  spike_data is (K, T, N); see the reference for the modes."""
    keep = torch.ones(spike_data.shape, dtype=torch.int64, device=spike_data.device)

    def regions_of(region):
        return np.argwhere(neuron_regions == region).flatten()

    if mode == 'manual':                                   # given neurons
        held = heldout_idxs
        keep[:, :, held] = 0
    elif mode == 'most':                                   # the n_active most active neurons
        rate = spike_data.detach().cpu().numpy().mean(axis=(0, 1))
        held = np.array(np.argsort(rate)[-n_active:])
        keep[:, :, held] = 0
    elif mode == 'inter_region':                           # hide whole regions, score the chosen neurons of each
        chosen = []
        for region in target_regions:
            idxs = regions_of(region)
            keep[:, :, idxs] = 0
            chosen.append(idxs[heldout_idxs])
        held = np.stack(chosen).flatten()
    elif mode == 'intra_region':                           # only the target regions are visible, minus the chosen neurons
        keep.zero_()
        chosen = []
        for region in target_regions:
            idxs = regions_of(region)
            keep[:, :, idxs] = 1
            if len(heldout_idxs) == 0:
                chosen.append(idxs)
            else:
                keep[:, :, idxs[heldout_idxs]] = 0
                chosen.append(idxs[heldout_idxs])
        held = np.stack(chosen).flatten()
    elif mode in ('forward_pred', 'modal_spike'):          # given time bins
        held = heldout_idxs
        keep[:, held, :] = 0
    elif mode == 'modal_behavior':
        held = heldout_idxs
        keep[:, held] = 0
    else:
        raise NotImplementedError('mode not implemented')
    return {"spikes": spike_data * keep, "heldout_idxs": held, "eval_mask": 1 - keep}


def _to_cuda(a):
    t = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a)
    if not torch.cuda.is_available():
        raise RuntimeError("the spike-prediction metrics run on the MI355X (csrc/metrics.hip); no CPU path in this package")
    return t.to("cuda", non_blocking=True)


def bits_per_spike(rates, spikes):
    """utils/eval_utils.py:1095-1119, on the device; accepts numpy arrays like upstream or tensors."""
    from multi_modal_foundation_model_amd.metrics import bits_per_spike as _bps
    return _bps(_to_cuda(rates), _to_cuda(spikes))


def bits_per_spike_per_neuron(rates, spikes):
    """The `for n_i in range(N): bits_per_spike(preds[:, :, [n_i]], gt[:, :, [n_i]])` loop of
    spiking_activity_recon_eval (utils/eval_utils.py:846-851) in one pass; returns a numpy array, inf -> nan."""
    from multi_modal_foundation_model_amd.metrics import bits_per_spike_per_neuron as _bpsn
    return _bpsn(_to_cuda(rates), _to_cuda(spikes)).cpu().numpy()

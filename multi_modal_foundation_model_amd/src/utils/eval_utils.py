"""The slice of the reference's `utils/eval_utils.py` that sits next to the hot path: the held-out
mask construction of the evaluation modes and the spike-prediction metrics.  Plotting, PSTH analysis and the dataset-bound
drivers (`load_model_data_local`, the HuggingFace dataset access and the figure code of `co_smoothing_eval`) are out of scope;
the computational core of `co_smoothing_eval` is `co_smoothing_core` below.

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


# ------------------------------------------------------------------------------------------------ forward-only evaluation driver
_MASK_MODE = {"per_neuron": "neuron", "forward_pred": "causal", "modal_spike": "causal", "inter_region": "inter-region",
              "intra_region": "intra-region", "modal_behavior": "causal"}


def eval_mod_dict(model, batch, mask_result, mask_mode, use_mtm=False, masked_mod='ap'):
    """The `mod_dict` the reference's evaluation loops build (utils/eval_utils.py:157-193): unmasked inputs (or the
    held-out-zeroed spikes with use_mtm), targets = the spikes, eval_mask = the held-out mask; behaviour passes through.
    masked_mod = 'behavior' is the `modal_behavior` loop (:638-700): the held-out mask sits on the behaviour channels and the
    spikes pass through with an all-zero eval_mask."""
    dev = batch['spikes_data'].device
    md = {}
    for mod in model.mod_to_indx.keys():
        d = dict(inputs_modality=torch.tensor(model.mod_to_indx[mod], device=dev), targets_modality=torch.tensor(model.mod_to_indx[mod], device=dev),
                 inputs_attn_mask=batch['time_attn_mask'], inputs_timestamp=batch['spikes_timestamps'],
                 targets_timestamp=batch['spikes_timestamps'], eid=batch['eid'][0] if 'eid' in batch else None,
                 num_neuron=batch['spikes_data'].shape[2], masking_mode=model.masker.mode if use_mtm else None)
        if mod == 'ap':
            d['inputs'] = (mask_result['spikes'] if use_mtm and masked_mod == 'ap' else batch['spikes_data']).clone()
            d['inputs_regions'] = batch.get('neuron_regions')
            d['targets'] = batch['spikes_data'].clone()
            d['eval_mask'] = mask_result['eval_mask'] if masked_mod == 'ap' else torch.zeros_like(batch['spikes_data']).to(torch.int64)
            d['mask_mode'] = mask_mode
        else:
            d['inputs'] = (mask_result['spikes'] if use_mtm and masked_mod == mod else batch['target']).clone()
            d['targets'] = batch['target'].clone()
            d['eval_mask'] = mask_result['eval_mask'] if masked_mod == mod else torch.zeros_like(batch['target']).to(torch.int64)
        md[mod] = d
    return md


def co_smoothing_core(model, batch, mode, heldout_idxs=None, target_regions=None, region_list=None, n_neurons=None, use_mtm=False):
    """One evaluation pass of `co_smoothing_eval` (utils/eval_utils.py:93-757) without the dataset / plotting code around it:
    held-out mask -> forward-only engine plan at B = len(test set) (`model.eval()`, `no_grad`) -> rates = exp(preds) on the device
    -> bits/spike of every scored neuron on its held-out slice (one launch of mmfm_bits_per_spike_neurons instead of the
    reference's per-neuron host loop) and the per-trial R^2 of those neurons (mmfm_r2_series).

    mode: 'per_neuron' (heldout_idxs = the ONE neuron to hide), 'forward_pred' / 'modal_spike' (heldout_idxs = time bins),
    'inter_region' / 'intra_region' (target_regions + heldout_idxs within each region; region_list = region of every neuron),
    'modal_behavior' (heldout_idxs = time bins of the BEHAVIOUR channels, :638-741: scored on the behaviour predictions as they are -
    no exp, bits/spike undefined = nan like upstream - so "gt" / "rates" are [K, T, n_beh] there and "neurons" the channels).
    Returns {"gt", "rates": [K, T, N] device tensors, "neurons": scored neuron indices, "bins": scored time bins,
    "bps": numpy [len(neurons)] (inf -> nan like upstream), "r2": numpy [len(neurons)] trial-averaged R^2, "loss"}."""
    from multi_modal_foundation_model_amd.metrics import bits_per_spike_per_neuron, r2_series
    spikes = batch['spikes_data']
    K_, T, N_all = spikes.shape
    N = N_all if n_neurons is None else n_neurons
    hd = np.array([] if heldout_idxs is None else heldout_idxs, dtype=np.int64)
    if mode == 'per_neuron':
        mask_result = heldout_mask(spikes.clone(), mode='manual', heldout_idxs=hd)
        neurons, bins = hd, np.arange(T)
    elif mode in ('forward_pred', 'modal_spike'):
        mask_result = heldout_mask(spikes.clone(), mode=mode, heldout_idxs=hd, target_regions=None, neuron_regions=region_list)
        neurons, bins = np.arange(N), hd
    elif mode in ('inter_region', 'intra_region'):
        mask_result = heldout_mask(spikes.clone(), mode=mode, heldout_idxs=hd, target_regions=target_regions, neuron_regions=region_list)
        neurons, bins = np.asarray(mask_result['heldout_idxs'], dtype=np.int64), np.arange(T)
    elif mode == 'modal_behavior':
        N = batch['target'].shape[2] if n_neurons is None else n_neurons
        mask_result = heldout_mask(batch['target'].clone(), mode=mode, heldout_idxs=hd, target_regions=None, neuron_regions=region_list)
        neurons, bins = np.arange(N), hd
    else:
        raise NotImplementedError(f"co_smoothing_core: mode {mode!r}")
    scored = 'behavior' if mode == 'modal_behavior' else 'ap'
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            out = model(eval_mod_dict(model, batch, mask_result, _MASK_MODE[mode], use_mtm=use_mtm, masked_mod=scored))
    finally:
        model.train(was_training)
    gt = out.mod_targets[scored][:, :, :N]
    rates = out.mod_preds[scored][:, :, :N] if scored == 'behavior' else torch.exp(out.mod_preds['ap'][:, :, :N])
    n_idx = torch.as_tensor(neurons, device=gt.device)
    t_idx = torch.as_tensor(bins, device=gt.device)
    g_sel = gt.index_select(1, t_idx).index_select(2, n_idx).contiguous()
    r_sel = rates.index_select(1, t_idx).index_select(2, n_idx).contiguous()
    bps = np.full(len(neurons), np.nan) if scored == 'behavior' else bits_per_spike_per_neuron(r_sel, g_sel).cpu().numpy()
    r2 = r2_series(g_sel, r_sel).double().cpu().numpy()                     # [K, n]: R^2 over the scored bins, per trial and neuron
    r2 = np.asarray([np.ma.masked_invalid(r2[:, j]).mean() for j in range(r2.shape[1])], dtype=np.float64)
    return dict(gt=gt, rates=rates, neurons=neurons, bins=bins, bps=bps, r2=r2, loss=out.loss)

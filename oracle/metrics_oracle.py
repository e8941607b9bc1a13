"""CPU oracle for the evaluation metrics.

TEST INFRASTRUCTURE ONLY (see oracle/mm_oracle.py).  Parity status:
* `neg_log_likelihood`, `bits_per_spike`: PINNED against tests/golden/eval_metrics.npz, produced by the reference's own
  functions (utils/eval_utils.py:1051-1119; pure numpy + scipy.special.gammaln).
* `r2`, `trial_avg_r2`: the reference wraps torcheval's R2Score (utils/metric_utils.py:2-11), and torcheval is not
  installed here nor on the GPU box -> PARITY UNPINNED for the R2Score call itself.  The restatement follows the formula
  torcheval documents (1 - SS_res / SS_tot on 1-D inputs) and the loop / masking structure of utils/utils.py:107-115,
  and is cross-checked against scikit-learn's r2_score (same published formula) in tests/test_metrics.py.
"""
import numpy as np
from scipy.special import gammaln


def neg_log_likelihood(rates, spikes):
    """utils/eval_utils.py:1051-1092 (NaN-free inputs): sum(r - n log r + log n!), zero rates -> 1e-9."""
    rates = np.array(rates, dtype=np.float64, copy=True)
    spikes = np.asarray(spikes, dtype=np.float64)
    assert spikes.shape == rates.shape
    assert np.all(rates >= 0)
    rates[rates == 0] = 1e-9
    return np.sum(rates - spikes * np.log(rates) + gammaln(spikes + 1.0))


def bits_per_spike(rates, spikes):
    """utils/eval_utils.py:1095-1119."""
    spikes = np.asarray(spikes, dtype=np.float64)
    nll_model = neg_log_likelihood(rates, spikes)
    null = np.tile(np.nanmean(spikes, axis=tuple(range(spikes.ndim - 1)), keepdims=True), spikes.shape[:-1] + (1,))
    nll_null = neg_log_likelihood(null, spikes)
    return (nll_null - nll_model) / np.nansum(spikes) / np.log(2)


def r2(y_true, y_pred):
    """R2Score on two 1-D series: 1 - sum (y - p)^2 / sum (y - mean y)^2 (float64)."""
    y, p = np.asarray(y_true, dtype=np.float64).ravel(), np.asarray(y_pred, dtype=np.float64).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        return 1.0 - np.sum((y - p) ** 2) / np.sum((y - y.mean()) ** 2)


def r2_series(gt, pred):
    """[G, S, C] -> [G, C]: R^2 of every (g, c) series over S."""
    gt, pred = np.asarray(gt), np.asarray(pred)
    return np.array([[r2(gt[g, :, c], pred[g, :, c]) for c in range(gt.shape[2])] for g in range(gt.shape[0])])


def trial_avg_r2(gt, pred):
    """utils/utils.py:109-115: for i in gt: r2 of every row of gt[i].T, invalid-masked mean; then the mean over i."""
    return float(np.mean([np.ma.masked_invalid(v).mean() for v in r2_series(gt, pred)]))


def bits_per_spike_per_neuron(rates, spikes):
    """spiking_activity_recon_eval's loop (utils/eval_utils.py:846-851): bits_per_spike on each neuron's [..., [n]] slice,
    inf -> nan."""
    rates, spikes = np.asarray(rates, dtype=np.float64), np.asarray(spikes, dtype=np.float64)
    out = np.empty(rates.shape[-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        for n in range(rates.shape[-1]):
            b = bits_per_spike(rates[..., [n]], spikes[..., [n]])
            out[n] = np.nan if np.isinf(b) else b
    return out

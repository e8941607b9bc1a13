"""Small host helpers used by the trainer (reference: utils/utils.py:20-36,107-132).  The
matplotlib analysis plots of the reference are out of scope; two minimal figure helpers keep
`MultiModalTrainer.plot_epoch` callable."""
import os
import random

import numpy as np
import torch

from utils.metric_utils import r2_score


def set_seed(seed):
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    print("seed set to {}".format(seed))


def move_batch_to_device(batch, device):
    for key, val in batch.items():
        if isinstance(val, torch.Tensor):
            batch[key] = val.to(device, non_blocking=True)
    return batch


def metrics_list(gt, pred, metrics=("r2",), device="cpu"):
    out = {}
    if "r2" in metrics and isinstance(gt, torch.Tensor) and gt.is_cuda and gt.dim() == 3:
        # one launch of mmfm_r2_series instead of gt.shape[0] x gt.shape[2] host-synchronised R2Score calls
        from multi_modal_foundation_model_amd.metrics import trial_avg_r2
        out["r2"] = trial_avg_r2(gt, pred.to(gt.device))
    elif "r2" in metrics:                   # mean over trials of the (nan-masked) mean over channels
        per_trial = []
        for i in range(gt.shape[0]):
            g, p = gt[i].T, pred[i].T
            vals = np.asarray([r2_score(g[k], p[k], device=device) for k in range(len(g))], dtype=np.float64)
            per_trial.append(np.ma.masked_invalid(vals).mean())
        out["r2"] = np.mean(per_trial)
    if "rsquared" in metrics:
        out["rsquared"] = np.mean([r2_score(gt[i], pred[i], device=device) for i in range(gt.shape[0])])
    if "mse" in metrics:
        out["mse"] = torch.mean((gt - pred) ** 2)
    if "mae" in metrics:
        out["mae"] = torch.mean(torch.abs(gt - pred))
    return out


def _pyplot():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def plot_gt_pred(gt, pred, epoch=0, modality="behavior"):
    plt = _pyplot()
    fig, axes = plt.subplots(1, 2, figsize=(12, 5))
    for ax, data, title in zip(axes, (gt, pred), ("Ground Truth", "Prediction")):
        ax.imshow(data, aspect="auto", cmap="binary" if modality == "ap" else "viridis")
        ax.set_title(f"{title} ({modality}), epoch {epoch}")
    return fig


def plot_neurons_r2(gt, pred, epoch=0, neuron_idx=()):
    plt = _pyplot()
    idx = list(neuron_idx)
    fig, axes = plt.subplots(max(1, len(idx)), 1, figsize=(12, 3 * max(1, len(idx))), squeeze=False)
    for ax, n in zip(axes[:, 0], idx):
        g, p = gt[:, n], pred[:, n]
        ax.plot(g.cpu().numpy(), label="gt")
        ax.plot(p.detach().cpu().numpy(), label="pred")
        ax.set_title(f"channel {n}, epoch {epoch}, r2 {r2_score(g, p):.3f}")
        ax.legend()
    return fig

"""R^2 on device without torcheval (reference: utils/metric_utils.py:4-11 wraps torcheval R2Score,
i.e. 1 - SS_res / SS_tot for 1-D inputs)."""
import torch


def r2_score(y_true, y_pred, device="cpu"):
    y_true = y_true.to(device).double().flatten()
    y_pred = y_pred.to(device).double().flatten()
    ss_res = ((y_true - y_pred) ** 2).sum()
    ss_tot = ((y_true - y_true.mean()) ** 2).sum()
    return (1.0 - ss_res / ss_tot).item()

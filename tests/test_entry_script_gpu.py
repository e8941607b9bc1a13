"""The drop-in entry point end to end: `src/train_multi_modal.py` (mirror of the reference's script, synthetic session)
trains for two short epochs on the MI355X, evaluates, plots and writes the whole-module checkpoints the reference's eval
scripts load (`torch.load(path)['model']`, utils/eval_utils.py:62)."""
import glob
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
SCRIPT = os.path.join(ROOT, "multi_modal_foundation_model_amd", "src", "train_multi_modal.py")


@pytest.mark.parametrize("dtype", ["bf16"])
def test_train_multi_modal_script_runs_and_checkpoints(tmp_path, dtype):
    cmd = [sys.executable, SCRIPT, "--mixed_training", "--epochs", "2", "--batches_per_epoch", "3", "--n_neurons", "64", "--dtype", dtype,
           "--base_path", str(tmp_path), "--overwrite"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    ckpts = glob.glob(os.path.join(str(tmp_path), "results", "**", "model_*.pt"), recursive=True)
    assert any(os.path.basename(c) == "model_last.pt" for c in ckpts), ckpts
    # the checkpoint is a whole-module pickle written by this repository's own code (weights_only=False is required to
    # rebuild the module; nothing from the reference is unpickled here)
    for p in (os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    last = [c for c in ckpts if c.endswith("model_last.pt")][0]
    blob = torch.load(last, weights_only=False, map_location="cpu")
    model = blob["model"]
    assert type(model).__name__ == "MultiModal" and blob["epoch"] >= 1
    sd = model.state_dict()
    assert len(sd) == 254 - 0 and all(torch.isfinite(v).all() for v in sd.values() if v.dtype.is_floating_point)

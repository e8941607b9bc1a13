"""Every masking mode of the API mirror's Masker against fixtures produced by the reference's Masker
(oracle/make_goldens.py:fx_masker_modes; models/masker.py:56-168): corrupted spikes and target masks are compared
bit for bit for two consecutive calls, and the torch CPU generator and Python `random` states afterwards must match
(= same draws in the same order).  SURVEY.md §8 row f4."""
import random

import numpy as np
import pytest
import torch

from conftest import load_npz


def _cases():
    z, meta = load_npz("masker_modes.npz")
    return z, meta


@pytest.mark.parametrize("cid", range(10))
def test_masker_mode_bit_exact_vs_reference_fixture(cid):
    from models.masker import Masker
    from utils.config_utils import DictConfig
    z, meta = _cases()
    case = meta[cid]
    mk = Masker(DictConfig(case["cfg"]))
    mk.train()
    torch.manual_seed(3 + cid)
    random.seed(17 + cid)
    ap = torch.from_numpy(z[f"c{cid}/ap"])
    regions = np.asarray([case["regions"]] * ap.shape[0])
    out1, m1 = mk(ap.clone(), regions)
    out2, m2 = mk(ap.clone(), regions)
    np.testing.assert_array_equal(m1.numpy(), z[f"c{cid}/mask1"], err_msg=f"{case['cfg']['mode']} mask (call 1)")
    np.testing.assert_array_equal(out1.numpy(), z[f"c{cid}/out1"], err_msg=f"{case['cfg']['mode']} spikes (call 1)")
    np.testing.assert_array_equal(m2.numpy(), z[f"c{cid}/mask2"], err_msg=f"{case['cfg']['mode']} mask (call 2)")
    np.testing.assert_array_equal(out2.numpy(), z[f"c{cid}/out2"], err_msg=f"{case['cfg']['mode']} spikes (call 2)")
    np.testing.assert_array_equal(torch.rand(3).numpy(), z[f"c{cid}/after"])
    assert random.random() == case["after_random"]
    assert m1.dtype == torch.int64 and m1.shape == ap.shape


def test_fixture_covers_every_mode():
    _, meta = _cases()
    assert {c["cfg"]["mode"] for c in meta} == {"temporal", "neuron", "random", "co-smooth", "forward-pred", "inter-region", "intra-region", "causal"}

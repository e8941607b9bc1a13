"""The N > 1 path with the REAL engine: two ranks (processes) on the one GPU of the test box, gloo for the collective (RCCL
needs one GPU per rank; the driver's 8-GPU run exercises it).  Checks what tests/test_ddp_cpu.py checks on a stand-in, end
to end: hipGraph-replayed backward segments with the bucketed all-reduce issued between them, parameters broadcast from
rank 0, and after three optimiser steps both replicas equal the single-process emulation that averages the per-rank
gradients of the per-rank normalised losses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
STEPS, WORLD, B, T, N_AP, N_BEH = 3, 2, 4, 8, 12, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (ROOT, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _batch(rank, step):
    from oracle import mm_oracle as O
    md = O.make_mod_dict(O.synth_batch(B, T, N_AP, N_BEH, seed=100 * rank + step), ("encoding", "decoding", "encoding")[step])
    for d in md.values():
        for k, v in list(d.items()):
            if isinstance(v, torch.Tensor):
                d[k] = v.cuda()
        d["targets_modality"], d["targets_timestamp"] = d["inputs_modality"], d["inputs_timestamp"]
    return md


def _make_model(kind, seed):
    """kind "tiny": fp32 parity mode, H = 32.  kind "bf16_fused": d_model 256, 2 + 2 layers, bf16 throughput mode with EVERY row-owner
    fused group on (MMFM_FUSED=15, what bench.py runs on the 8-GPU node): LayerNorm / linear gradients then come out of
    mmfm_ln_linear_grad inside the backward segments whose completion fires the bucket all-reduces."""
    from helpers import build_model, model_config, tiny_config
    if kind == "tiny":
        return build_model(tiny_config(n_enc=2, n_dec=2), N_AP, N_BEH, seed=seed).cuda().train()
    assert os.environ.get("MMFM_FUSED") == "15"        # set by the caller: the worker for its process, the test through monkeypatch
    model = build_model(model_config(n_enc=2, n_dec=2, dropout=0.0, emb_dropout=0.0), N_AP, N_BEH, seed=seed)
    model.compute_dtype = "bf16"
    return model.cuda().train()


def _worker(rank, port, out_dir, kind="tiny"):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK="0")
    if kind == "bf16_fused":
        os.environ["MMFM_FUSED"] = "15"
    import torch.distributed as dist
    from helpers import make_optimizer
    from multi_modal_foundation_model_amd.ddp import DataParallelModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    model = _make_model(kind, 7 + rank)                              # replicas differ until the broadcast
    ddp = DataParallelModel(model, bucket_bytes=(16 << 10) if kind == "tiny" else (2 << 20))   # several collectives per backward
    opt, sch = make_optimizer(ddp, 10)
    losses = []
    for s in range(STEPS):
        out = ddp(_batch(rank, s))
        out.loss.backward()
        opt.step(); sch.step(); opt.zero_grad()
        losses.append(out.loss.item())
    assert len(ddp._ddp.buckets.buckets) >= 3
    torch.save(dict(state={k: v.detach().cpu() for k, v in model.state_dict().items()}, losses=losses), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tiny", "bf16_fused"])
def test_two_ranks_match_the_gradient_averaging_emulation(tmp_path, kind, monkeypatch):
    _paths()
    if kind == "bf16_fused":
        monkeypatch.setenv("MMFM_FUSED", "15")
    mp.spawn(_worker, args=(_free_port(), str(tmp_path), kind), nprocs=WORLD, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=True)
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), f"replicas diverged: {k}"
    # emulation in this process: rank 0's initial parameters, per-rank gradients averaged, one optimiser step per step
    from helpers import make_optimizer
    model = _make_model(kind, 7)
    opt, sch = make_optimizer(model, 10)
    for s in range(STEPS):
        grads, losses = None, []
        for rank in range(WORLD):
            opt.zero_grad()
            out = model(_batch(rank, s))
            out.loss.backward()
            losses.append(out.loss.item())
            g = model._engine.G.clone()
            grads = g if grads is None else grads + g
        model._engine.G.copy_(grads / WORLD)
        opt.step(); sch.step()
        assert losses[0] == pytest.approx(r0["losses"][s], rel=1e-5) and losses[1] == pytest.approx(r1["losses"][s], rel=1e-5), s
    opt.zero_grad()
    if kind == "bf16_fused":
        assert model._engine._fused_mask(B * 2 * T) == 15
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.detach().cpu().numpy(), r0["state"][k].numpy(), rtol=2e-5, atol=2e-7, err_msg=k)


def _rccl_worker(rank, port, out_dir):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from helpers import build_model, make_optimizer, tiny_config
    from multi_modal_foundation_model_amd.ddp import DataParallelModel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)         # RCCL, as bench.py / Accelerator initialise it
    finals = []
    for wrap in (True, False):
        model = build_model(tiny_config(n_enc=2, n_dec=2), N_AP, N_BEH, seed=7).cuda().train()
        m = DataParallelModel(model, bucket_bytes=16 << 10) if wrap else model
        opt, sch = make_optimizer(m, 10)
        for s in range(STEPS):
            out = m(_batch(0, s))
            out.loss.backward()
            opt.step(); sch.step(); opt.zero_grad()
        if wrap:
            assert m._ddp.avg_op is not None and len(m._ddp.buckets.buckets) >= 3
        finals.append({k: v.detach().cpu() for k, v in model.state_dict().items()})
    torch.save(finals, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_rccl_collective_path_single_rank(tmp_path):
    """RCCL needs one GPU per rank, so with one GPU only world_size 1 can run it: the AVG all-reduce is then an identity,
    but the calls are the ones the 8-GPU run makes (nccl backend bound to the device, async bucket all-reduces issued between
    the hipGraph-replayed backward segments, stream-side wait before the optimiser).  Result must equal the unwrapped run."""
    _paths()
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    with_ddp, without = torch.load(os.path.join(tmp_path, "rccl.pt"), weights_only=True)
    for k in without:
        assert torch.equal(with_ddp[k], without[k]), k

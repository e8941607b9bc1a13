"""N>1 path on CPU (gloo, world_size 2): the bucketed gradient all-reduce of the DDP wrapper.

The engine itself needs the GPU, so these tests drive `EngineDDP` through a stand-in that has the
engine's flat-buffer interface (layout, G, hooks) and takes its per-rank gradients from the CPU
oracle.  What is checked is the wrapper's contract: after backward every rank holds
the MEAN over ranks of the per-rank gradients (per-rank normalised losses), bucket by bucket, in the
order backward completes them, and the replicas' parameters are identical after the broadcast.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from helpers import tiny_config


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeEngine:
    """The slice of Engine that EngineDDP uses, on CPU tensors."""

    def __init__(self, layout, cfg):
        self.layout, self.cfg = layout, cfg
        self.P = torch.zeros(layout.n)
        self.G = torch.zeros(layout.n)
        self.grad_ready_hooks, self.backward_done_hooks = [], []
        self.dtype = "fp32"

    def refresh_weights(self):
        pass

    def run_backward(self, grads_by_name):
        """Write gradients segment by segment in backward order, firing the hooks like Engine.backward."""
        from multi_modal_foundation_model_amd.ddp import backward_order
        for seg in backward_order(self.layout, self.cfg):
            s, e = next((s, e) for n, s, e in self.layout.segments if n == seg)
            for name, (off, shape) in self.layout.entries.items():
                if s <= off < e:
                    self.layout.view(self.G, name).copy_(grads_by_name[name])
            for h in self.grad_ready_hooks:
                h(seg)
        for h in self.backward_done_hooks:
            h()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_modal_foundation_model_amd.ddp import EngineDDP
    from multi_modal_foundation_model_amd.engine import EngineConfig, ParamLayout
    from oracle import mm_oracle as O
    mc = tiny_config(n_enc=2, n_dec=2)
    ec = EngineConfig.from_model_config(mc, [("ap", 12), ("behavior", 2)])
    lay = ParamLayout(ec)
    eng = FakeEngine(lay, ec)
    # replicas start different on purpose: the wrapper must broadcast rank 0's parameters
    ocfg = O.OracleCfg.from_model_config(mc, {"ap": 12, "behavior": 2})
    sd = O.init_state_dict(ocfg, seed=100 + rank)
    for k in O.trainable_keys(sd, ocfg):
        lay.view(eng.P, k).copy_(sd[k])
    ddp = EngineDDP(eng, bucket_bytes=16 << 10)          # small buckets -> several collectives
    assert len(ddp.buckets.buckets) >= 3
    # per-rank batch (different data and masks per rank, same objective), per-rank normalised loss
    sd = {k: lay.view(eng.P, k).clone() for k in O.trainable_keys(sd, ocfg)}
    sd = O.share_mod_emb(sd, ocfg)
    keys = O.trainable_keys(sd, ocfg)
    for k in keys:
        sd[k].requires_grad_(True)
    mk = O.OracleMasker(dict(force_active=True, mode="temporal", ratio=0.3, zero_ratio=1.0, random_ratio=1.0, expand_prob=0.0,
                             max_timespan=1, channels=None, timesteps=None, mask_regions=["all"], target_regions=["all"],
                             n_mask_regions=1, causal_zero=True))
    torch.manual_seed(7 + rank)
    out = O.forward(sd, O.make_mod_dict(O.synth_batch(3, 8, 12, 2, seed=1000 * rank), "token_masking"), ocfg, training=True, masker=mk)
    grads = dict(zip(keys, torch.autograd.grad(out["loss"], [sd[k] for k in keys])))
    eng.run_backward(grads)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=eng.P.numpy(), G=eng.G.numpy(),
             local=torch.cat([grads[k].flatten() for k in keys]).numpy(), n=float(sum(out["mod_n_examples"].values())))
    dist.barrier()
    dist.destroy_process_group()


def test_engine_ddp_bucketed_allreduce_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    np.testing.assert_array_equal(r[0]["P"], r[1]["P"])                     # replicas identical after the broadcast
    np.testing.assert_array_equal(r[0]["G"], r[1]["G"])                     # every rank holds the same reduced gradient
    assert r[0]["n"] != r[1]["n"]                                            # per-rank masks differ -> per-rank normalisation
    # the reduced gradient is the mean of the per-rank gradients (single-process emulation)
    sys.path.insert(0, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"))
    from multi_modal_foundation_model_amd.engine import EngineConfig, ParamLayout
    from oracle import mm_oracle as O
    mc = tiny_config(n_enc=2, n_dec=2)
    lay = ParamLayout(EngineConfig.from_model_config(mc, [("ap", 12), ("behavior", 2)]))
    ocfg = O.OracleCfg.from_model_config(mc, {"ap": 12, "behavior": 2})
    keys = O.trainable_keys(O.init_state_dict(ocfg, seed=0), ocfg)
    mean = (r[0]["local"] + r[1]["local"]) / 2
    off = 0
    G = torch.from_numpy(r[0]["G"])
    for k in keys:
        v = lay.view(G, k).numpy().ravel()
        np.testing.assert_allclose(v, mean[off:off + v.size], rtol=1e-6, atol=1e-9, err_msg=k)
        off += v.size
    assert off == mean.size


def test_accelerator_single_process_is_passthrough():
    from multi_modal_foundation_model_amd.ddp import Accelerator
    os.environ.pop("WORLD_SIZE", None)
    acc = Accelerator()
    m = torch.nn.Linear(2, 2)
    assert acc.prepare(m) is m or isinstance(acc.prepare(m), torch.nn.Linear)
    assert acc.is_main_process


def _save_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"))
    from trainer.base import MultiModalTrainer
    t = MultiModalTrainer.__new__(MultiModalTrainer)              # the checkpoint methods only: no model build, no GPU
    torch.manual_seed(100 + rank)                                  # per-rank RNG streams, as under data parallelism
    t.model = torch.nn.Linear(3, 2)
    t.optimizer = torch.optim.AdamW(t.model.parameters(), lr=1e-3)
    t.lr_scheduler, t.log_dir, t.session_active_neurons = None, out_dir, []
    t.save_model(name="last", epoch=3)
    dist.barrier()
    dist.destroy_process_group()


def test_checkpoint_writes_are_rank_guarded_gloo(tmp_path):
    """Under data parallelism the module pickle is written by rank 0 only (replicas are identical; concurrent writers tear the file) and
    every rank keeps its own training state (its RNG streams differ): trainer/base.py save_model / save_train_state."""
    world, port = 2, _free_port()
    mp.spawn(_save_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    files = sorted(os.listdir(tmp_path))
    assert files == ["model_last.pt", "train_state_last_rank0.pt", "train_state_last_rank1.pt"], files
    st = [torch.load(tmp_path / f"train_state_last_rank{r}.pt", weights_only=False) for r in range(world)]
    assert st[0]["epoch"] == st[1]["epoch"] == 3
    assert not torch.equal(st[0]["torch_rng"], st[1]["torch_rng"])           # each rank's own stream


def _best_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "multi_modal_foundation_model_amd", "src"))
    from trainer.base import MultiModalTrainer
    from utils.config_utils import DictConfig

    class Acc:
        device = torch.device("cpu")
    t = MultiModalTrainer.__new__(MultiModalTrainer)              # the epoch loop with stand-in epochs: no model build, no GPU
    t.model = torch.nn.Linear(3, 2)
    t.optimizer = torch.optim.AdamW(t.model.parameters(), lr=1e-3)
    t.lr_scheduler, t.log_dir, t.session_active_neurons, t.accelerator = None, out_dir, [], Acc()
    t.config = DictConfig(dict(training=dict(num_epochs=3, save_plot_every_n_epochs=1000)))
    t.metric, t.use_wandb, t.modal_filter = "r2", False, {"output": []}
    # rank-local evaluation shards: the metric improves on rank 1 only in epoch 1 and on rank 0 only in epoch 2
    metric = {0: [0.1, 0.1, 0.3], 1: [0.1, 0.2, 0.2]}[rank]
    t.train_epoch = lambda epoch: {"train_loss": 0.0}
    state = {"e": -1}

    def eval_epoch():
        state["e"] += 1
        return {"eval_loss": 1.0, "eval_trial_avg_r2": metric[state["e"]], "eval_gt": {}, "eval_preds": {}}
    t.eval_epoch = eval_epoch
    saved = []
    real_save = t.save_model
    t.save_model = lambda name="last", epoch=0: (saved.append((name, epoch)), real_save(name=name, epoch=epoch))[1]
    t.train()                                                      # deadlocks (mp.spawn join never returns) if ranks disagree on "best"
    with open(os.path.join(out_dir, f"saved{rank}.txt"), "w") as f:
        f.write(repr(saved))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_best_checkpoint_decision_is_collective_gloo(tmp_path):
    """`save_model` ends in a barrier, and `train()` calls it for 'best' from a comparison of RANK-LOCAL
    evaluation metrics.  Rank 0's comparison decides for every rank; with per-rank decisions one rank would wait in the barrier
    while the other went on to the next epoch's gradient all-reduce."""
    world, port = 2, _free_port()
    mp.spawn(_best_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s0, s1 = (open(tmp_path / f"saved{r}.txt").read() for r in range(world))
    assert s0 == s1 == repr([("best", 0), ("best", 2), ("last", 2)]), (s0, s1)      # rank 0's metric: improves in epochs 0 and 2

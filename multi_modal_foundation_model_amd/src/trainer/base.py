"""`MultiModalTrainer` — the reference's epoch loop (trainer/base.py:10-308) around the HIP engine.

Same constructor kwargs (log_dir, accelerator, lr_scheduler, config, num_neurons, avail_mod,
modal_filter, mixed_training), same methods and return values.  Differences, all on the host side:
  * the per-step `loss.item()` sync (trainer/base.py:199) is deferred to the end of the epoch: losses
    stay on the device and are summed in float64 in step order, which gives the same `train_loss`;
  * wandb / matplotlib are optional (absent in this image): logging falls back to print / PNG files.
"""
import os
import random

import numpy as np
import torch

from utils.utils import metrics_list, move_batch_to_device, plot_gt_pred, plot_neurons_r2

try:                       # optional, like the reference's `config.wandb.use`
    import wandb
except Exception:          # pragma: no cover
    wandb = None

OBJECTIVES = ['encoding', 'decoding', 'token_masking']


class LazyRegions:
    """The [B, N] region array `np.asarray(batch['neuron_regions']).T` of trainer/base.py:57, built on first use.

    Converting the collated list of N lists of B strings costs 34 ms per step at B = 1024 (as much host time as the whole
    GPU step), and on this path nobody reads it after the masker's first call: the model accepts temporal masking only
    (mm.py:66) and the masker touches the regions once, to expand 'all' (masker.py:72-76).  Behaves like the ndarray for
    readers: `np.asarray(x)`, `np.unique(x)`, `x == region`, indexing, `.shape`."""
    __slots__ = ("_raw", "_arr")

    def __init__(self, raw):
        self._raw, self._arr = raw, None

    def materialize(self):
        if self._arr is None:
            self._arr = np.asarray(self._raw).T
        return self._arr

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None else a.astype(dtype)

    def __eq__(self, other):
        return self.materialize() == other

    def __ne__(self, other):
        return self.materialize() != other

    __hash__ = None

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    @property
    def shape(self):
        return self.materialize().shape

    @property
    def T(self):
        return self.materialize().T


class MultiModalTrainer():
    def __init__(self, model, train_dataloader, eval_dataloader, optimizer, **kwargs):
        self.model = model
        self.train_dataloader = train_dataloader
        self.eval_dataloader = eval_dataloader
        self.optimizer = optimizer

        self.log_dir = kwargs.get("log_dir", None)
        self.accelerator = kwargs.get("accelerator", None)
        self.lr_scheduler = kwargs.get("lr_scheduler", None)
        self.config = kwargs.get("config", None)
        self.num_neurons = kwargs.get("num_neurons", None)

        self.model_class = self.config.model.model_class
        self.metric = 'r2'
        self.session_active_neurons = []
        self.avail_mod = kwargs.get("avail_mod", None)
        self.modal_filter = kwargs.get("modal_filter", None)
        self.mod_to_indx = {r: i for i, r in enumerate(self.avail_mod)}

        if self.config.training.mask_type == "input":
            self.masking_schemes = self.config.training.mask_mode
        else:
            self.masking_mode = None
        self.mixed_training = kwargs.get("mixed_training", False)
        if self.mixed_training:
            self.training_schemes = list(OBJECTIVES)
        else:
            self.training_mode = None
        self.use_wandb = bool(self.config.wandb.use) and wandb is not None
        # mask_type 'embd': the model keeps only the token-level mask of the masker and DISCARDS the corrupted spikes
        # (mm.py:262-267), so the masker's three full-size [B, T, N] host draws are dead work (330 ms per step at B = 1024, ten GPU
        # steps).  The trainer therefore switches the masker to its token-mask-only stream - identically distributed masks, a
        # shorter walk through the host generator - unless `training.exact_masker_stream: true` (or MMFM_EXACT_MASKER=1) asks for
        # the reference's exact generator stream, which the parity tests do.
        masker = getattr(self.model, "masker", None)
        exact = bool(self.config.training.get("exact_masker_stream", False)) if hasattr(self.config.training, "get") else False
        exact = exact or os.environ.get("MMFM_EXACT_MASKER", "0") == "1"
        if masker is not None and hasattr(masker, "token_mask_only"):
            # a per-run switch: set on every construction (never inherited from a loaded checkpoint, Masker.__getstate__ drops it)
            masker.token_mask_only = bool(self.config.training.mask_type == "embd" and not exact)
            if self._rank_world()[0] == 0:
                print("(train) masker stream: " + (
                    "token-mask-only (same mask distribution; from the second masker call on NOT the reference's generator stream - "
                    "set training.exact_masker_stream: true or MMFM_EXACT_MASKER=1 for bit-exact masks)" if masker.token_mask_only
                    else "reference-exact (every [B,T,N] corruption draw of models/masker.py is taken)"))
        # host-side constants of the batch -> mod_dict translation, built once instead of every step: the
        # modality-index scalars (a pageable H2D copy each = a stream drain per call) and the [B, N] region array
        self._mod_index_cache = {}
        self._const_mask_cache = {}

    # ------------------------------------------------------------------ batch -> mod_dict (trainer/base.py:51-103)
    def _forward_model_outputs(self, batch, masking_mode, training_mode):
        single_modal = len(self.modal_filter['output']) == 1
        dev = self.accelerator.device
        batch = move_batch_to_device(batch, dev)
        spikes, behav = batch['spikes_data'], batch['target']

        def const_mask(like, value):
            # upstream materialises torch.ones_like(x).to(int64) / zeros (547 MB per mask at B=1024, up to four per step);
            # the model only ever reads mask[:, :, 0] (mm.py:270), so an expanded view of one cached element is equivalent
            key = (value, str(like.device))
            if key not in self._const_mask_cache:
                self._const_mask_cache[key] = torch.full((1, 1, 1), value, dtype=torch.int64, device=like.device)
            return self._const_mask_cache[key].expand(like.shape)

        mod_dict = {}
        for mod, idx in self.mod_to_indx.items():
            key = (mod, str(dev))
            if key not in self._mod_index_cache:
                self._mod_index_cache[key] = torch.tensor(idx, device=dev)
            idx_t = self._mod_index_cache[key]
            d = {
                'inputs_modality': idx_t, 'targets_modality': idx_t,
                'inputs_attn_mask': batch['time_attn_mask'], 'inputs_timestamp': batch['spikes_timestamps'],
                'targets_timestamp': batch['spikes_timestamps'], 'eid': batch['eid'][0],
                'num_neuron': spikes.shape[2], 'masking_mode': masking_mode,
            }
            if mod == 'ap':
                # (upstream clones twice; nothing on this path writes into either tensor - the masker works on its own
                # clone and the engine copies inputs/targets into its static buffers - so the 2 x 274 MB copies are dropped)
                d['inputs'], d['targets'] = spikes, spikes
                d['inputs_regions'] = LazyRegions(batch['neuron_regions'])
            elif mod == 'behavior':
                d['inputs'], d['targets'] = behav, behav
            else:
                raise Exception(f"Modality not implemented yet.")
            d['eval_mask'] = const_mask(spikes, 1 if (single_modal and mod in self.modal_filter['output']) else 0)
            mod_dict[mod] = d

        if not single_modal:
            if training_mode == 'encoding':        # predict spikes from behaviour: every ap bin is a target
                for mod in mod_dict:
                    mod_dict[mod]['eval_mask'] = const_mask(spikes, 1 if mod == 'ap' else 0)
            elif training_mode == 'decoding':      # predict behaviour from spikes
                for mod in mod_dict:
                    mod_dict[mod]['eval_mask'] = const_mask(behav, 1 if mod == 'behavior' else 0)
            elif training_mode == 'token_masking':  # the model's masker draws the targets
                for mod in mod_dict:
                    mod_dict[mod]['eval_mask'] = None
            else:
                raise Exception(f"Training objective not implemented yet.")
        return self.model(mod_dict)

    # ------------------------------------------------------------------ epochs
    def train(self):
        best_eval_loss = torch.tensor(float('inf'))
        best_metric = -torch.tensor(float('inf'))
        epoch = -1
        for epoch in range(self.config.training.num_epochs):
            train_res = self.train_epoch(epoch)
            eval_res = self.eval_epoch()
            print(f"epoch: {epoch} train loss: {train_res['train_loss']}")
            key = f'eval_trial_avg_{self.metric}'
            if eval_res:
                # every rank evaluates its own shard, so "improved" can differ between ranks while save_model is collective
                # (rank-guarded write + barrier): rank 0 decides for all, or one rank would sit in the barrier while the
                # others wait for it in the next step's gradient all-reduce
                if self._rank0_decides(bool(eval_res[key] > best_metric)):
                    best_eval_loss, best_metric = eval_res['eval_loss'], eval_res[key]
                    print(f"epoch: {epoch} best eval loss: {best_eval_loss} trial avg {self.metric}: {best_metric}")
                    self.save_model(name="best", epoch=epoch)
                    self._log_figures(eval_res, epoch, prefix="best_")
                print(f"epoch: {epoch} eval loss: {eval_res['eval_loss']} trial avg {self.metric}: {eval_res[key]}")
            if epoch % self.config.training.save_plot_every_n_epochs == 0:
                self._log_figures(eval_res, epoch, prefix="")
            if self.use_wandb:
                wandb.log({"train_loss": train_res['train_loss'], "eval_loss": eval_res['eval_loss'], key: eval_res[key]})
        self.save_model(name="last", epoch=epoch)
        if self.use_wandb:
            wandb.log({"best_eval_loss": best_eval_loss, f"best_eval_trial_avg_{self.metric}": best_metric})

    def _log_figures(self, eval_res, epoch, prefix):
        for mod in self.modal_filter['output']:
            try:
                figs = self.plot_epoch(gt=eval_res['eval_gt'][0][mod], preds=eval_res['eval_preds'][0][mod], epoch=epoch,
                                       active_neurons=self.session_active_neurons[0][:5], modality=mod)
            except ImportError:          # matplotlib not installed: plotting is optional
                return
            if self.use_wandb:
                wandb.log({f"{prefix}gt_pred_fig_{mod}": wandb.Image(figs['plot_gt_pred']),
                           f"{prefix}r2_fig_{mod}": wandb.Image(figs['plot_r2'])})
            else:
                figs['plot_gt_pred'].savefig(os.path.join(self.log_dir, f"{prefix}gt_pred_fig_{mod}_{epoch}.png"))
                figs['plot_r2'].savefig(os.path.join(self.log_dir, f"{prefix}r2_fig_{mod}_{epoch}.png"))

    def _sample_modes(self):
        if self.config.training.mask_type == "input":
            self.masking_mode = random.sample(self.masking_schemes, 1)[0]
        if self.mixed_training:
            self.training_mode = random.sample(self.training_schemes, 1)[0]

    def train_epoch(self, epoch):
        self.model.train()
        losses = []
        for batch in self.train_dataloader:
            self._sample_modes()
            outputs = self._forward_model_outputs(batch, masking_mode=self.masking_mode, training_mode=self.training_mode)
            loss = outputs.loss
            loss.backward()
            self.optimizer.step()
            self.lr_scheduler.step()
            self.optimizer.zero_grad()
            losses.append(loss.detach())
        # one host sync per epoch instead of one per step; float64 sum in step order == sum of .item()s
        train_loss = float(sum(x.item() for x in torch.stack(losses).double().cpu())) if losses else 0.
        return {"train_loss": train_loss}

    def eval_epoch(self):
        self.model.eval()
        eval_loss = 0.
        results = {n: {mod: {"gt": [], "preds": []} for mod in self.modal_filter['output']} for n in self.num_neurons}
        if not self.eval_dataloader:
            return None
        with torch.no_grad():
            for batch in self.eval_dataloader:
                self._sample_modes()
                outputs = self._forward_model_outputs(batch, masking_mode=self.masking_mode, training_mode=self.training_mode)
                eval_loss += outputs.loss.item()
                n = batch['spikes_data'].shape[2]
                for mod in self.modal_filter['output']:
                    sl = slice(None, n) if mod == 'ap' else slice(None)
                    results[n][mod]["gt"].append(outputs.mod_targets[mod].clone()[:, :, sl])
                    results[n][mod]["preds"].append(outputs.mod_preds[mod].clone()[:, :, sl])
        gt, preds, scores = {}, {}, []
        for idx, n in enumerate(self.num_neurons):
            gt[idx], preds[idx] = {}, {}
            for mod in self.modal_filter['output']:
                g = torch.cat(results[n][mod]["gt"], dim=0)
                p = torch.cat(results[n][mod]["preds"], dim=0)
                gt[idx][mod] = g
                preds[idx][mod] = torch.exp(p) if mod == 'ap' else p        # rates from log-rates
            for mod in self.modal_filter['output']:
                active = np.argsort(gt[idx][mod].cpu().numpy().sum((0, 1)))[::-1][:50].tolist()
                self.session_active_neurons.append(active)
                if mod == 'ap':
                    sel = self.session_active_neurons[idx]
                    res = metrics_list(gt=gt[idx][mod][:, :, sel].transpose(-1, 0), pred=preds[idx][mod][:, :, sel].transpose(-1, 0),
                                       metrics=["r2"], device=self.accelerator.device)
                else:
                    res = metrics_list(gt=gt[idx][mod], pred=preds[idx][mod], metrics=[self.metric], device=self.accelerator.device)
                scores.append(res[self.metric])
        return {"eval_loss": eval_loss, f"eval_trial_avg_{self.metric}": np.nanmean(scores), "eval_gt": gt, "eval_preds": preds}

    def plot_epoch(self, gt, preds, epoch, active_neurons, modality):
        fig = plot_gt_pred(gt=gt.mean(0).T.cpu().numpy(), pred=preds.mean(0).T.detach().cpu().numpy(), epoch=epoch, modality=modality)
        if modality == 'behavior':
            active_neurons = range(gt.size()[-1])
        r2_fig = plot_neurons_r2(gt=gt.mean(0), pred=preds.mean(0), neuron_idx=active_neurons, epoch=epoch)
        return {"plot_gt_pred": fig, "plot_r2": r2_fig}

    def save_model(self, name="last", epoch=0):
        """Whole-module pickle like the reference (trainer/base.py:302-308) so its eval scripts can load it; next to it the
        training state the reference never saves (optimiser moments + step, OneCycleLR, dropout / masker / objective RNG) so
        that a run can RESUME: `load_train_state`."""
        rank, world = self._rank_world()
        model = getattr(self.model, "module", self.model)
        if rank == 0:                 # replicas are identical: one writer, or concurrent ranks tear the file
            print(f"saving model: {name} to {self.log_dir}")
            torch.save({"model": model, "epoch": epoch}, os.path.join(self.log_dir, f"model_{name}.pt"))
        self.save_train_state(name=name, epoch=epoch)
        if world > 1:
            torch.distributed.barrier()

    def _rank0_decides(self, flag):
        """Rank 0's boolean on every rank (one tiny broadcast; a no-op without data parallelism)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return bool(flag)
        dev = self.accelerator.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.broadcast(t, src=0)
        return bool(int(t.item()))

    @staticmethod
    def _rank_world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def train_state_path(self, name="last", rank=None):
        """One file per rank under data parallelism: optimiser state is replicated, but the RNG streams (dropout counter, masker
        generator, python / numpy objective sampling) are per rank, and a resumed rank needs its own."""
        rank = self._rank_world()[0] if rank is None else rank
        world = self._rank_world()[1]
        return os.path.join(self.log_dir, f"train_state_{name}.pt" if world == 1 else f"train_state_{name}_rank{rank}.pt")

    def save_train_state(self, name="last", epoch=0):
        model = getattr(self.model, "module", self.model)
        eng = getattr(model, "_engine", None)
        state = dict(epoch=epoch, optimizer=self.optimizer.state_dict(),
                     lr_scheduler=self.lr_scheduler.state_dict() if self.lr_scheduler is not None else None,
                     engine_rng=None if eng is None else eng.rng.detach().cpu().clone(),
                     engine_dtype=None if eng is None else eng.dtype,
                     torch_rng=torch.get_rng_state(), python_rng=random.getstate(), numpy_rng=np.random.get_state(),
                     masker=dict(mode=model.masker.mode, ratio=model.masker.ratio, mask_regions=model.masker.mask_regions,
                                 target_regions=model.masker.target_regions) if hasattr(model, "masker") else None,
                     session_active_neurons=list(self.session_active_neurons))
        torch.save(state, self.train_state_path(name))

    def load_train_state(self, path=None, name="last"):
        """Restore what `save_train_state` wrote into THIS trainer (model parameters come from the module pickle or a
        state_dict the caller loaded; the engine is created on the spot so the flat buffers exist).  Returns the epoch."""
        # weights_only=False: the file holds python / numpy RNG tuples next to the tensors - only ever a file this trainer wrote
        state = torch.load(path or self.train_state_path(name), weights_only=False)
        model = getattr(self.model, "module", self.model)
        eng = model.engine()
        self.optimizer.load_state_dict(state["optimizer"])
        if self.lr_scheduler is not None and state["lr_scheduler"] is not None:
            self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        if state["engine_rng"] is not None:
            eng.rng.copy_(state["engine_rng"].to(eng.rng.device))
        torch.set_rng_state(state["torch_rng"])
        random.setstate(state["python_rng"])
        np.random.set_state(state["numpy_rng"])
        if state.get("masker") and hasattr(model, "masker"):
            for k, v in state["masker"].items():
                setattr(model.masker, k, v)
        self.session_active_neurons = list(state.get("session_active_neurons", []))
        return state["epoch"]

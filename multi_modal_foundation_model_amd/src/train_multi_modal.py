"""Entry script: masked pretraining of the spike + behaviour MultiModal model on an MI355X.

Same flags, config files, model/optimiser/scheduler/trainer construction order as the reference's
`src/train_multi_modal.py`; the session data comes from the synthetic loader because the HuggingFace
datasets (`neurofm123/<eid>_aligned`, reference lines 97-119) cannot be downloaded offline.  Any
iterable yielding the loader batch dict (loader/base.py:436-450) can be passed instead.

    cd multi_modal_foundation_model_amd && python src/train_multi_modal.py --mixed_training --epochs 2
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 src/train_multi_modal.py --mixed_training
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(os.path.dirname(HERE))):
    if p not in sys.path:
        sys.path.insert(0, p)
os.chdir(os.path.dirname(HERE))           # config paths are 'src/configs/...' like upstream

import torch
from torch.optim.lr_scheduler import OneCycleLR

from multi_modal.decoder_embeddings import DecoderEmbedding
from multi_modal.encoder_embeddings import EncoderEmbedding
from multi_modal.mm import MultiModal
from multi_modal_foundation_model_amd.ddp import Accelerator
from multi_modal_foundation_model_amd.optim import make_optimizer
from multi_modal_foundation_model_amd.synthetic import SyntheticLoader
from trainer.make import make_multimodal_trainer
from utils.config_utils import config_from_kwargs, update_config
from utils.utils import set_seed

ap = argparse.ArgumentParser()
ap.add_argument("--eid", type=str, default="synthetic-session")
ap.add_argument("--mask_ratio", type=float, default=0.1)
ap.add_argument("--mask_mode", type=str, default="temporal")
ap.add_argument("--use_MtM", action="store_true")
ap.add_argument("--mixed_training", action="store_true")
ap.add_argument("--overwrite", action="store_true")
ap.add_argument("--base_path", type=str, default="/tmp/mmfm_results")
ap.add_argument("--epochs", type=int, default=None, help="override training.num_epochs (2000 upstream)")
ap.add_argument("--batches_per_epoch", type=int, default=8)
ap.add_argument("--n_neurons", type=int, default=668)
ap.add_argument("--dtype", type=str, default=os.environ.get("MMFM_DTYPE", "fp32"), choices=["fp32", "bf16"])
args = ap.parse_args()

avail_beh = ["wheel-speed", "whisker-motion-energy"]
config = config_from_kwargs({"model": "include:src/configs/multi_modal/mm.yaml"})
config = update_config("src/configs/multi_modal/trainer_mm.yaml", config)
config["model"]["masker"]["mode"] = args.mask_mode
config["model"]["masker"]["ratio"] = args.mask_ratio
if args.epochs is not None:
    config["training"]["num_epochs"] = args.epochs
set_seed(config.seed)

avail_mod = ["ap", "behavior"]
modal_filter = {"input": ["ap", "behavior"], "output": ["ap", "behavior"]}
mask_mode = "-".join(config.training.mask_mode) if config.training.mask_type == "input" else args.mask_mode
log_dir = os.path.join(args.base_path, "results", f"ses-{args.eid}", "set-train", f"inModal-{'-'.join(modal_filter['input'])}",
                       f"outModal-{'-'.join(modal_filter['output'])}", f"mask-{config.training.mask_type}", f"mode-{mask_mode}",
                       f"ratio-{args.mask_ratio}", f"mixedTraining-{args.mixed_training}")
assert not os.path.exists(os.path.join(log_dir, "model_last.pt")) or args.overwrite, "last checkpoint exists and overwrite is False"
os.makedirs(log_dir, exist_ok=True)

accelerator = Accelerator()
n_behaviors, n_neurons = len(avail_beh), args.n_neurons
meta_data = {"num_neurons": [n_neurons], "eids": [args.eid]}
T = config.data.max_time_length
train_dataloader = SyntheticLoader(args.batches_per_epoch, config.training.train_batch_size, T, n_neurons, n_behaviors,
                                   rank=accelerator.rank, seed0=0)
val_dataloader = SyntheticLoader(2, config.training.test_batch_size, T, n_neurons, n_behaviors, rank=accelerator.rank, seed0=10 ** 6)

encoder_embeddings, decoder_embeddings = {}, {}
for mod in modal_filter["input"]:
    encoder_embeddings[mod] = EncoderEmbedding(hidden_size=config.model.encoder.transformer.hidden_size,
                                               n_channel=n_neurons if mod == "ap" else n_behaviors, config=config.model.encoder)
for mod in modal_filter["output"]:
    decoder_embeddings[mod] = DecoderEmbedding(hidden_size=config.model.decoder.transformer.hidden_size,
                                               n_channel=n_neurons if mod == "ap" else n_behaviors,
                                               output_channel=n_neurons if mod == "ap" else n_behaviors, config=config.model.decoder)

model = MultiModal(encoder_embeddings, decoder_embeddings, avail_mod=avail_mod, config=config.model,
                   share_modality_embeddings=True, **config.method.model_kwargs, **meta_data)
model.compute_dtype = args.dtype
print("(train) masking mode: ", model.masker.mode)
print("(train) masking ratio: ", model.masker.ratio)
print("(train) masking active: ", model.masker.force_active)
model = accelerator.prepare(model)

optimizer = make_optimizer(model, lr=config.optimizer.lr, weight_decay=config.optimizer.wd, eps=config.optimizer.eps)
lr_scheduler = OneCycleLR(optimizer=optimizer,
                          total_steps=config.training.num_epochs * len(train_dataloader) // config.optimizer.gradient_accumulation_steps,
                          max_lr=config.optimizer.lr, pct_start=config.optimizer.warmup_pct, div_factor=config.optimizer.div_factor)

trainer_ = make_multimodal_trainer(model=model, train_dataloader=train_dataloader, eval_dataloader=val_dataloader,
                                   optimizer=optimizer, log_dir=log_dir, accelerator=accelerator, lr_scheduler=lr_scheduler,
                                   avail_mod=avail_mod, modal_filter=modal_filter, mixed_training=args.mixed_training, config=config,
                                   **meta_data)
trainer_.train()

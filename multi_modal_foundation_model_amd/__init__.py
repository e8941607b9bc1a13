"""MI355X-native masked-pretraining path (hand-written HIP kernels behind a C-ABI).

`multi_modal_foundation_model_amd.src/` mirrors the reference's Python surface
(`multi_modal.mm.MultiModal`, `trainer.make.make_multimodal_trainer`, ...) so that
`src/train_multi_modal.py` is a drop-in; the device work runs in `libmmfm_hip.so`.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(PKG_DIR, "src")
__version__ = "0.1.0"

"""ctypes binding of libmmfm_hip.so (the C-ABI declared in include/mmfm.h).

There is NO fallback: if the shared library is missing or a call fails this raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C multi_modal_foundation_model_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import re

# torch ships its own ROCm runtime (libamdhip64 under torch/lib).  It MUST be loaded before
# libmmfm_hip.so so that both resolve to the same HIP runtime instance: device pointers and streams
# handed over by torch are only meaningful inside the runtime that created them.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMFM_LIB") or os.path.join(_HERE, "libmmfm_hip.so")    # MMFM_LIB: A/B another build of the same library
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mmfm.h")

F32, BF16 = 0, 1
ATTN_DIAG, ATTN_CAUSAL, ATTN_SEP = 1, 2, 4
ACT_NONE, ACT_GELU, ACT_SOFTSIGN, ACT_GELU_GRAD, ACT_SOFTSIGN_GRAD, ACT_SOFTSIGN_GRAD_OUT = 0, 1, 2, 3, 4, 5


class MmfmError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [("state", C.c_void_p), ("site", C.c_uint32), ("p", C.c_float)]


NO_DROP = Dropout(None, 0, 0.0)


class GemmDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("c_f32", C.c_int), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("a_kcontig", C.c_int), ("b_kcontig", C.c_int), ("splits", C.c_int), ("kchunk", C.c_int),
                ("slab_stride", C.c_int64), ("bias", C.c_void_p), ("pre_out", C.c_void_p), ("act", C.c_int),
                ("act_scale", C.c_float), ("gradmul_pre", C.c_void_p), ("drop", Dropout), ("residual", C.c_void_p),
                ("ldr", C.c_int), ("colsum", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int), ("dh", C.c_int),
                ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("ldq", C.c_int), ("ldk", C.c_int),
                ("ldv", C.c_int), ("o", C.c_void_p), ("ldo", C.c_int), ("lse", C.c_void_p), ("keypad", C.c_void_p),
                ("mod_id", C.c_void_p), ("flags", C.c_int), ("scale", C.c_float), ("drop_p", Dropout),
                ("drop_o", Dropout), ("d_o", C.c_void_p), ("lddo", C.c_int), ("dq", C.c_void_p), ("dk", C.c_void_p),
                ("dv", C.c_void_p), ("lddq", C.c_int), ("lddk", C.c_int), ("lddv", C.c_int), ("keepbits", C.c_void_p)]


class PrepEntry(C.Structure):
    _fields_ = [("W", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("bias", C.c_void_p), ("Wp", C.c_void_p),
                ("WpT", C.c_void_p), ("bp", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("tile0", C.c_int), ("pad_", C.c_int),
                ("WpP", C.c_void_p), ("WpTP", C.c_void_p)]


class ReduceEntry(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("n", C.c_int64), ("slab_stride", C.c_int64), ("nslabs", C.c_int32),
                ("accumulate", C.c_int32), ("chunk0", C.c_int32), ("pad_", C.c_int32)]


class RowGemmDesc(C.Structure):
    _fields_ = [("R", C.c_int64), ("K", C.c_int), ("N", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int), ("w", C.c_void_p),
                ("ldw", C.c_int), ("bias", C.c_void_p), ("ln", C.c_int), ("eps", C.c_float), ("xhat", C.c_void_p),
                ("rstd", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int), ("y", C.c_void_p), ("ldy", C.c_int),
                ("stream_out", C.c_int), ("rotate", C.c_int), ("ln_bwd", C.c_int), ("bwd_xhat", C.c_void_p), ("bwd_rstd", C.c_void_p)]


class MlpDesc(C.Structure):
    _fields_ = [("R", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int), ("eps", C.c_float), ("w_up", C.c_void_p),
                ("b_up", C.c_void_p), ("w_down", C.c_void_p), ("b_down", C.c_void_p), ("drop", Dropout), ("y", C.c_void_p),
                ("ldy", C.c_int), ("xhat", C.c_void_p), ("rstd", C.c_void_p), ("dy", C.c_void_p), ("lddy", C.c_int),
                ("w_down_t", C.c_void_p), ("w_up_t", C.c_void_p), ("t1", C.c_void_p), ("g", C.c_void_p), ("du", C.c_void_p),
                ("dx", C.c_void_p), ("lddx", C.c_int), ("rotate", C.c_int)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PROTOS = {
    "mmfm_version": (C.c_int, []),
    "mmfm_last_error": (C.c_char_p, []),
    "mmfm_device_check": (C.c_int, [_i]),
    "mmfm_rng_seed": (C.c_int, [_vp, C.c_uint64, _vp]),
    "mmfm_rng_advance": (C.c_int, [_vp, _vp]),
    "mmfm_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "mmfm_gemm_pair": (C.c_int, [_vp, _vp, _vp]),
    "mmfm_gemm_dw_tiles": (C.c_int, [_i, _i, _i]),
    "mmfm_reduce_slabs": (C.c_int, [_vp, _vp, _i64, _i, _i64, _i, _vp]),
    "mmfm_reduce_slabs_multi": (C.c_int, [_vp, _i, _i, _vp]),
    "mmfm_colsum_workspace": (C.c_int64, [_i64, _i]),
    "mmfm_colsum": (C.c_int, [_i, _vp, _i64, _i, _i, _vp, _i, _vp, _i64, _vp]),
    "mmfm_layernorm_fwd": (C.c_int, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "mmfm_layernorm_bwd_workspace": (C.c_int64, [_i64, _i]),
    "mmfm_layernorm_bwd": (C.c_int, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "mmfm_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "mmfm_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "mmfm_attn_keepbits_bytes": (C.c_int64, [_i, _i, _i, _i]),
    "mmfm_attn_keep_prob": (C.c_float, [_f]),
    "mmfm_mask_prep": (C.c_int, [_i, _i, _i, C.POINTER(_vp), C.POINTER(_i64), _vp, C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmfm_collate_csr": (C.c_int, [_i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmfm_stitch_fwd": (C.c_int, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mmfm_stitch_bwd_workspace": (C.c_int64, [_i, _i, _i, _i, _i, _i]),
    "mmfm_stitch_bwd": (C.c_int, [_i, _vp, _vp, _vp, _vp, Dropout, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "mmfm_masked_loss_workspace": (C.c_int64, [_i64, _i]),
    "mmfm_masked_loss_fwd": (C.c_int, [_i, _i, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _i64, _vp]),
    "mmfm_loss_finalize": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mmfm_masked_loss_bwd": (C.c_int, [_i, _i, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "mmfm_dropout_apply": (C.c_int, [_i, _vp, _vp, _i64, _i, Dropout, _vp]),
    "mmfm_cast_f32_to_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mmfm_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mmfm_prep_weights": (C.c_int, [_vp, _i, _i, _vp]),
    "mmfm_rowgemm": (C.c_int, [C.POINTER(RowGemmDesc), _vp]),
    "mmfm_mlp_fwd": (C.c_int, [C.POINTER(MlpDesc), _vp]),
    "mmfm_mlp_bwd": (C.c_int, [C.POINTER(MlpDesc), _vp]),
    "mmfm_ln_linear_grad_workspace": (C.c_int64, [_i]),
    "mmfm_ln_linear_grad": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i64, _vp]),
    "mmfm_r2_series": (C.c_int, [_vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _i, _i, _i, _vp, _vp]),
    "mmfm_bits_per_spike_workspace": (C.c_int64, [_i64, _i]),
    "mmfm_bits_per_spike": (C.c_int, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "mmfm_bits_per_spike_neurons_workspace": (C.c_int64, [_i64, _i]),
    "mmfm_bits_per_spike_neurons": (C.c_int, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
}

_lib = None


def header_symbols():
    """Every function the C header declares (used by the CPU export test)."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmfm_[a-z0-9_]+)\s*\(", src)))


def lib():
    """The loaded library with typed prototypes.  Raises ImportError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU "
                              "fallback.  Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise MmfmError(f"{what} failed (code {rc}): {lib().mmfm_last_error().decode(errors='replace')}")

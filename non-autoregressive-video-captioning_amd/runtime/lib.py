"""ctypes binding of libnacf_hip.so (the C ABI declared in include/nacf_hip.h).

The product path has NO fallback: if the shared library is missing or an entry
point fails, this raises -- it never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("NACF_HIP_LIB") or os.path.join(_PKG_DIR, "libnacf_hip.so")   # env: tuning builds only

# activations (nacf_hip.h)
ACT_NONE, ACT_RELU, ACT_GELU_NEW, ACT_TANH, ACT_SIGMOID, ACT_TANH_SIGMOID, ACT_GELU_ERF = range(7)
ACT_BY_NAME = {"gelu_new": ACT_GELU_NEW, "gelu": ACT_GELU_ERF, "relu": ACT_RELU}


class Epilogue(ctypes.Structure):
    """struct nacf_epilogue."""
    _fields_ = [
        ("bias", c_void_p), ("act", c_int32), ("act_split", c_int32),
        ("preact", c_void_p), ("ld_preact", c_int64),
        ("p_drop1", c_float), ("salt1", c_uint32),
        ("residual", c_void_p), ("ld_residual", c_int64),
        ("p_drop2", c_float), ("salt2", c_uint32),
        ("row_tokens", c_void_p), ("rng_state", c_void_p),
    ]


class CritTail(ctypes.Structure):
    """struct nacf_crit_tail."""
    _fields_ = [("n_pass", c_int32), ("label_logp", c_void_p * 4), ("argmax", c_void_p * 4), ("labels", c_void_p * 4),
                ("rows", c_int32 * 4), ("exclude", c_int32 * 4), ("slot", c_int32 * 4),
                ("kl_x", c_void_p), ("kl_t", c_void_p), ("kl_total", c_int32), ("kl_slot", c_int32)]


class RowSet(ctypes.Structure):
    """struct nacf_rowset."""
    _fields_ = [("rows", c_void_p), ("count", c_void_p), ("zero_dead", ctypes.c_int32)]


class WImageDesc(ctypes.Structure):
    """struct nacf_wimage_desc: one weight matrix of the bf16 image table (nacf_wimage_refresh)."""
    _fields_ = [("w", c_void_p), ("img", c_void_p), ("imgT", c_void_p),
                ("ld", c_int64), ("plane", c_int64), ("planeT", c_int64),
                ("N", c_int32), ("K", c_int32), ("tile0", c_int32), ("tiles_k", c_int32)]


# GEMM arithmetic modes (nacf_hip.h NACF_GEMM_*)
GEMM_F32, GEMM_BF16, GEMM_BF16X3 = 0, 1, 3
GEMM_MODE_BY_NAME = {"f32": GEMM_F32, "bf16": GEMM_BF16, "bf16x3": GEMM_BF16X3}
GEMM_MODE_NAME = {v: k for k, v in GEMM_MODE_BY_NAME.items()}

_P, _I, _L, _F, _U, _S = c_void_p, c_int, c_int64, c_float, c_uint32, c_size_t
_EP = POINTER(Epilogue)
_RS = POINTER(RowSet)

# name -> (restype, argtypes); mirrors include/nacf_hip.h one to one
SIGNATURES = {
    "nacf_last_error": (c_char_p, []),
    "nacf_version": (c_int, []),
    "nacf_abi_count": (c_int, []),
    "nacf_rng_advance": (c_int, [_P, _P]),
    "nacf_rowset_build": (c_int, [_P, _P, _L, _P, _P, _P]),
    "nacf_linear_fwd": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _EP, _RS, _P]),
    "nacf_linear_bwd_data_workspace": (_S, [_I, _I, _I]),
    "nacf_linear_bwd_data": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _F, _P, _S, _RS, _P]),
    "nacf_linear_bwd_weight_workspace": (_S, [_I, _I, _I]),
    "nacf_gemm_config": (c_int, [_I, _I, _I, _I, _P, _P]),
    "nacf_gemm_set_mode": (c_int, [_I]),
    "nacf_gemm_get_mode": (c_int, []),
    "nacf_gemm_last_kernel": (c_char_p, []),
    "nacf_wimage_register": (c_int, [_P, _I, _I, _L, _P, _L, _P, _L, _I]),
    "nacf_wimage_unregister": (c_int, [_P, _L]),
    "nacf_wimage_refresh": (c_int, [_P, _I, _I, _I, _P]),
    "nacf_linear_bwd_weight": (c_int, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _F, _P, _S, _RS, _P]),
    "nacf_dw_group_begin": (c_int, [_I]),
    "nacf_dw_group_stats": (c_int, [_P, _P]),
    "nacf_dw_group_flush": (c_int, [_P]),
    "nacf_dw_group_launch_gemms": (c_int, [_P]),
    "nacf_dw_group_pending": (c_int, []),
    "nacf_wide_group_begin": (c_int, []),
    "nacf_wide_group_flush": (c_int, [_P]),
    "nacf_epilogue_bwd": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _EP, _RS, _P]),
    "nacf_sample_frames": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _U, _P, _P, _P, _P]),
    "nacf_gather_clips_h2d": (c_int, [_P, _P, _P, _I, ctypes.c_size_t, _P]),
    "nacf_gather_clips_zc": (c_int, [_P, _P, _P, _I, ctypes.c_size_t, _I, _P]),
    "nacf_build_targets": (c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_double, ctypes.c_double, _U, _P,
                                   _P, _P, _P, _P, _P]),
    "nacf_loss_combine": (c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "nacf_loss_combine_bwd": (c_int, [_P, _P, _I, _I, _P, _P]),
    "nacf_crit_tail_fwd": (c_int, [POINTER(CritTail), _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "nacf_crit_tail_bwd": (c_int, [POINTER(CritTail), _P, _P, _I, _I, _P, _P, _P]),
    "nacf_highway_mix_fwd": (c_int, [_P, _P, _P, _I, _I, _F, _U, _P, _P]),
    "nacf_highway_mix_bwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _F, _U, _P, _P]),
    "nacf_bn_workspace": (_S, [_I, _I]),
    "nacf_bn_concat_fwd": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _S, _P]),
    "nacf_bn_concat_bwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _S, _P]),
    "nacf_bn_sync_merge": (c_int, [_P, _I, _I, _I, _P, _L, _P, _P, _P]),
    "nacf_bn_concat_fwd_multi": (c_int, [_I, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _S, _P]),
    "nacf_bn_concat_bwd_multi": (c_int, [_I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _S, _P]),
    "nacf_bn_sync_local_multi": (c_int, [_I, _P, _I, _P, _I, _P, _P, _S, _P]),
    "nacf_bn_sync_bwd_local_multi": (c_int, [_I, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _S, _P]),
    "nacf_bn_sync_stat": (c_int, [_P, _I, _I, _P, _L, _P, _P, _S, _P]),
    "nacf_bn_concat_fwd_sync": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _L, _P]),
    "nacf_bn_sync_bwd_stat": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _S, _P]),
    "nacf_bn_concat_bwd_sync": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "nacf_mean_time_fwd": (c_int, [_P, _P, _I, _I, _I, _P]),
    "nacf_mean_time_bwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "nacf_log_softmax_rows": (c_int, [_P, _P, _I, _I, _P]),
    "nacf_log_softmax_rows_bwd": (c_int, [_P, _P, _P, _I, _I, _P]),
    "nacf_kldiv_mean": (c_int, [_P, _P, _P, _P, _P, _F, _I, _I, _P]),
    "nacf_embed_ln_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U, _P, _P]),
    "nacf_embed_ln_bwd_workspace": (_S, [_I, _I, _I]),
    "nacf_embed_ln_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _F, _U, _P, _P, _S, _P]),
    "nacf_embed_scatter_bwd_workspace": (_S, [_I, _I, _I, _I]),
    "nacf_embed_scatter_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _S, _P]),
    "nacf_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _I, _I, _F, _U, _P, _P, _P]),
    "nacf_layernorm_bwd_workspace": (_S, [_I, _I]),
    "nacf_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _F, _U, _P, _P, _P, _S, _P]),
    "nacf_attention_fwd": (c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nacf_attention_bwd": (c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I,
                                   _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nacf_attention_fwd_dropout": (c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U, _P, _P]),
    "nacf_attention_bwd_dropout": (c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I,
                                           _I, _I, _I, _I, _I, _I, _I, _I, _F, _U, _P, _P]),
    "nacf_masked_mean_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "nacf_vocab_logsoftmax_fwd": (c_int, [_P, _L, _I, _I, _P, _P, _P, _P, _I, _P]),
    "nacf_nll_reduce": (c_int, [_P, _P, _P, _I, _I, _P, _P]),
    "nacf_xent_bwd": (c_int, [_P, _L, _P, _L, _I, _I, _P, _P, _F, _I, _P]),
    "nacf_vocab_lse_fwd": (c_int, [_P, _L, _P, _L, _P, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P, _S, _RS, _P]),
    "nacf_xent_bwd_lse": (c_int, [_P, _L, _P, _P, _L, _I, _I, _P, _P, _F, _I, _P]),
    "nacf_xent_bwd_lse_multi": (c_int, [_P, _L, _P, _P, _L, _I, _I, _I, _P, _P, _F, _I, _P]),
    "nacf_nll_reduce_multi": (c_int, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "nacf_vocab_logsoftmax_bwd": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _P]),
    "nacf_vocab_argmax_workspace": (_S, [_I, _I]),
    "nacf_vocab_argmax": (c_int, [_P, _L, _P, _L, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _S, _RS, _P]),
    "nacf_length_beam": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "nacf_length_beam_gold": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "nacf_canvas_init": (c_int, [_P, _I, _I, _P, _P]),
    "nacf_canvas_init_gold": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "nacf_select_mask": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "nacf_token_replace": (c_int, [_P, _L, _L, _L, _P]),
    "nacf_teacher_probs": (c_int, [_P, _P, _P, _L, _P]),
    "nacf_init_probs": (c_int, [_P, _P, _L, _P]),
    "nacf_apply_mask": (c_int, [_P, _P, _L, _L, _P]),
    "nacf_mask_rank": (c_int, [_P, _I, _I, _P, _P, _P]),
    "nacf_select_rank": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "nacf_easy_first_update": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "nacf_best_candidate": (c_int, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P]),
    "nacf_beam_step": (c_int, [_P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nacf_adam_step": (c_int, [_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _F, _P]),
    "nacf_adam_step_part": (c_int, [_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _F, _I, _P]),
    "nacf_rmsprop_step": (c_int, [_P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _I, _P]),
}

_lib = None


class NacfLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen libnacf_hip.so and type every entry point.  Raises (never falls
    back) when the library has not been built: run `python __graft_entry__.py`
    or `make -C non-autoregressive-video-captioning_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NacfLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Build it with `make -C non-autoregressive-video-captioning_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise NacfLibraryError(f"{LIB_PATH} does not export {name}; rebuild the extension")
        fn.restype = res
        fn.argtypes = args
    built = lib.nacf_abi_count()
    if built != len(SIGNATURES):
        raise NacfLibraryError(f"{LIB_PATH} was built with {built} entry points, this package binds {len(SIGNATURES)}: "
                               "stale build, run `make -C non-autoregressive-video-captioning_amd/csrc`")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().nacf_last_error()
        raise NacfLibraryError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

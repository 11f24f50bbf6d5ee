"""ctypes binding of libemogest_hip.so (include/emogest.h).

The library is the product: there is no CPU or eager-PyTorch fallback anywhere in this package.
If the shared object is missing, importing any compute entry point raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libemogest_hip.so")

EG_PREC_F32, EG_PREC_BF16X3, EG_PREC_BF16 = 0, 1, 2
PRECISIONS = {"f32": EG_PREC_F32, "fp32": EG_PREC_F32, "bf16x3": EG_PREC_BF16X3, "bf16": EG_PREC_BF16}

(PACK_RAW, PACK_LINEAR, PACK_VEC_PAD, PACK_CONV3X3, PACK_BN_SCALE, PACK_BN_SHIFT, PACK_CONV1X1, PACK_STEM,
 PACK_WN_TAP, PACK_CONV1D, PACK_POS_TABLE, PACK_LINEAR_T, PACK_LINEAR_FOLD, PACK_BIAS_FOLD) = range(14)


class EgError(RuntimeError):
    pass


class EgWeightEntry(C.Structure):
    _fields_ = [("key", C.c_char * 192), ("kind", C.c_int32), ("dims", C.c_int32 * 4),
                ("offset", C.c_int64), ("numel", C.c_int64)]


class EgGeneratorConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "frames", "pose_dim", "prior_frames", "chunk", "d_model", "d_inner", "n_layers", "n_head", "d_k",
        "n_mels", "spec_len", "text_len", "n_words", "embed_dim", "tcn_hidden", "tcn_layers", "variant",
        "precision", "n_position")] + [("reserved", C.c_int32 * 5)]


class EgCvaeConfig(C.Structure):
    _fields_ = [("frames", C.c_int32), ("d_model", C.c_int32), ("latent", C.c_int32), ("n_classes", C.c_int32),
                ("reserved", C.c_int32 * 4)]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64


class EgLinearArgs(C.Structure):
    """include/emogest.h: the argument block of eg_linear_ex (field order = the header's)."""
    _fields_ = [(n, _P) for n in ("x", "w", "bias", "res1", "res2", "y", "gate_src", "drop_epoch", "partial", "x_images", "y_images")] + \
               [("drop_offset", C.c_uint64)] + \
               [(n, _I) for n in ("lda", "ldw", "ldr", "ldc", "ldg", "m", "n", "k", "relu", "precision", "splits", "k_x", "y_k")] + \
               [("drop_seed", C.c_uint32), ("drop_p", C.c_float)]


# name -> (restype, argtypes).  Must list every symbol include/emogest.h declares
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "eg_last_error": (C.c_char_p, []),
    "eg_launch_count": (_L, []),
    "eg_launch_histogram": (_L, [C.c_char_p, _L, C.c_int32]),
    "eg_version": (C.c_char_p, []),
    "eg_set_default_precision": (C.c_int, [C.c_int]),
    "eg_get_default_precision": (C.c_int, []),
    "eg_generator_default_config": (C.c_int, [C.POINTER(EgGeneratorConfig)]),
    "eg_generator_create": (C.c_int, [C.POINTER(EgGeneratorConfig), C.POINTER(_P)]),
    "eg_generator_destroy": (None, [_P]),
    "eg_generator_arena_floats": (_L, [_P]),
    "eg_generator_num_weights": (_I, [_P]),
    "eg_generator_weight_entry": (C.c_int, [_P, _I, C.POINTER(EgWeightEntry)]),
    "eg_generator_workspace_bytes": (_L, [_P, _I]),
    "eg_generator_forward": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "eg_generator_forward_draws": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "eg_generator_draws_workspace_bytes": (_L, [_P, _I, _I]),
    "eg_generator_tap": (C.c_int, [_P, _I, _P, C.c_char_p, C.POINTER(_P), C.POINTER(_L)]),
    "eg_cvae_default_config": (C.c_int, [C.POINTER(EgCvaeConfig)]),
    "eg_cvae_create": (C.c_int, [C.POINTER(EgCvaeConfig), C.POINTER(_P)]),
    "eg_cvae_destroy": (None, [_P]),
    "eg_cvae_arena_floats": (_L, [_P]),
    "eg_cvae_num_weights": (_I, [_P]),
    "eg_cvae_weight_entry": (C.c_int, [_P, _I, C.POINTER(EgWeightEntry)]),
    "eg_cvae_workspace_bytes": (_L, [_P, _I]),
    "eg_cvae_sample": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _L, _P]),
    "eg_cvae_forward": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "eg_mel_tables": (C.c_int, [_P, _P, _P, _P]),
    "eg_mel_workspace_bytes": (_L, [_I, _I]),
    "eg_melspectrogram": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _L, _P]),
    "eg_conv3x3": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_conv3x3_se": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_se_gate_pre": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_conv3x3_gap_tiles": (_I, [_I, _I, _I, _I, _I]),
    "eg_stem_conv": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_se_gate": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "eg_se_residual_relu": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_linear": (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_split_tiles": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "eg_linear_presplit": (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_linear_splitk": (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "eg_layernorm": (C.c_int, [_P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "eg_attention": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_attention_masked": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _L, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_mha_workspace_bytes": (_L, [_I, _I, _I, _I, _I]),
    "eg_multi_head_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_ffn_workspace_bytes": (_L, [_I, _I, _I]),
    "eg_positionwise_ffn": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P]),
    "eg_tcn_forward": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_add_rows": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "eg_conv3x3_packed_floats": (_L, [_I, _I]),
    "eg_profile_enable": (C.c_int, [_I]),
    "eg_profile_disable": (C.c_int, []),
    "eg_profile_read": (_I, [_P, _P, _P, _I]),
    "eg_profile_read_workgroups": (_I, [_P, _I]),
    "eg_reparameterize": (C.c_int, [_P, _P, _P, _P, _L, _P]),
    "eg_conv1d": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_contrastive_workspace_bytes": (C.c_int64, [_I]),
    "eg_contrastive_loss": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _L, _P]),
    "eg_contrastive_backward_workspace_bytes": (_L, [_I]),
    "eg_contrastive_loss_backward": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _L, _P]),
    # ---- training-path primitives (csrc/train.hip)
    "eg_transpose": (C.c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "eg_gemm_tn_workspace_floats": (_L, [_I, _I, _L]),
    "eg_gemm_tn": (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _L, _P, _L, _I, _P]),
    "eg_conv3x3_wgrad": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_im2col3x3": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_subsample": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_im2col1d": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_conv1d_cl_forward": (C.c_int, [_P, _P, _P, _P] + [_I] * 9 + [_P]),
    "eg_conv1d_cl_backward_input": (C.c_int, [_P, _P, _P, _P] + [_I] * 9 + [_P]),
    "eg_conv1d_cl_backward_weight_workspace_floats": (_L, [_I] * 7),
    "eg_conv1d_cl_backward_weight": (C.c_int, [_P, _P, _P, _P, _P] + [_I] * 9 + [_P, _L, _P]),
    "eg_pad_cols": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "eg_colreduce_workspace_floats": (_L, [_I]),
    "eg_bn_train_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "eg_bn_train_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "eg_bn_train_forward_gap": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "eg_bn_train_forward_sq": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "eg_conv3x3_sq": (C.c_int, [_P, _P, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "eg_se_gate_train_forward": (C.c_int, [_P] * 12 + [_I, _I, _I, _P]),
    "eg_se_tail_forward": (C.c_int, [_P] * 9 + [_I, _I, _I, _P]),
    "eg_se_tail_backward_reduce": (C.c_int, [_P] * 7 + [_I, _I, _I, _P, _P]),
    "eg_se_gate_train_backward": (C.c_int, [_P] * 16 + [_I, _I, _I, _P]),
    "eg_se_tail_backward_finish": (C.c_int, [_P] * 14 + [_I, _I, _I, _P]),
    "eg_se_tail_backward_apply": (C.c_int, [_P] * 13 + [_I, _I, _I, _P]),
    "eg_colsum": (C.c_int, [_P, _P, _P, _P, _L, _I, _P, _P]),
    "eg_elementwise": (C.c_int, [_P, _P, _P, _L, _I, C.c_float, _P]),
    "eg_conv3x3_wgrad_mfma_workspace_floats": (_L, [_I, _I, _I, _I, _I]),
    "eg_conv3x3_wgrad_mfma": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_conv3x3_wgrad_mfma_oihw": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_conv3x3_wgrad_gather_mfma": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_conv3x3_dgrad_s2": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_conv3x3_res_masked": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_conv3x3_wgrad_mfma_oihw_in_affine": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "eg_conv3x3_sq_in_affine": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eg_bn_train_stats_sq": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "eg_pack_conv3x3_device": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "eg_linear_packed_floats": (_L, [_I, _I]),
    "eg_pack_linear_device": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "eg_pack_table_blocks": (_I, [_I, _I, _I, _I]),
    "eg_pack_table": (C.c_int, [_P, _I, _I, _P]),
    "eg_dropout": (C.c_int, [_P, _P, _L, C.c_float, C.c_uint32, C.c_uint64, _P]),
    "eg_dropout_dev": (C.c_int, [_P, _P, _L, C.c_float, C.c_uint32, C.c_uint64, _P, _P]),
    "eg_seg_mean": (C.c_int, [_P, _P, _I, _I, _I, C.c_float, _P, _P]),
    "eg_seg_dot": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "eg_se_scale": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "eg_layernorm_backward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "eg_attention_backward": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "eg_attention_train": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, C.c_float, C.c_uint32, C.c_uint64, _P, _P]),
    "eg_attention_backward_train": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, C.c_float, C.c_uint32,
                                              C.c_uint64, _P, _P]),
    "eg_smooth_l1": (C.c_int, [_P, _P, _P, _P, _L, C.c_float, C.c_float, _P, _P]),
    "eg_cross_entropy": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, _P, _P, _I, _I, _P, _P]),
    "eg_kld": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "eg_adam_step": (C.c_int, [_P, _P, _P, _P, _L, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I, _P]),
    "eg_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _L, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "eg_counter_add": (C.c_int, [_P, _I, _P]),
    "eg_sp_gate_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_sp_gate_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_tm_scale_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_tm_scale_backward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eg_linear_wgrad_mfma_workspace_floats": (_L, [_I, _I, _I]),
    "eg_linear_wgrad_mfma": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _L, _P]),
    "eg_linear_ex": (C.c_int, [C.POINTER(EgLinearArgs), _P]),
    "eg_layernorm_backward_ex_workspace_floats": (_L, [_I, _I]),
    "eg_layernorm_backward_ex": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, C.c_float, C.c_float, C.c_uint32, C.c_uint64, _P, _P, _P, _P]),
    "eg_layernorm_img": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "eg_f32_to_bf16": (C.c_int, [_P, _P, _L, _P]),
    "eg_bf16_to_f32": (C.c_int, [_P, _P, _L, C.c_float, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises EgError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64; loading ours before it would pull a second HIP runtime copy
    # from /opt/rocm into the process ("no ROCm-capable device" on the first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EgError(
            f"{LIB_PATH} not found: the HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "emotiongestures_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().eg_last_error()
        raise EgError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def precision_code(p) -> int:
    if isinstance(p, int):
        return p
    try:
        return PRECISIONS[str(p).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {p!r}; choose from {sorted(PRECISIONS)}")

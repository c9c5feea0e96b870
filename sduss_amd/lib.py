"""ctypes binding of libmxdenoise.so (the C ABI declared in include/mxdenoise.h).

The product path has no fallback: if the HIP library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmxdenoise.so")

MX_F32, MX_F16, MX_BF16 = 0, 1, 2
EPI_SILU, EPI_GEGLU, EPI_OUT_F32, EPI_QKV, EPI_GELU_TANH, EPI_RES_BCAST, EPI_RMSNORM, EPI_GELU, EPI_QUICK_GELU, EPI_GEGLU_TANH = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512


class MxError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference's launcher only printed
    cudaGetLastError and went on -- norm_silu_concat.cu:434-437; SURVEY.md section 8b asks to raise)."""


MAX_SEGS = 4


class GemmSeg(C.Structure):
    """mx_gemm_seg: one problem of a grouped launch (include/mxdenoise.h)"""
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p), ("c", C.c_void_p), ("residual", C.c_void_p), ("vt", C.c_void_p), ("rowbias", C.c_void_p),
        ("gate", C.c_void_p), ("ln_stats", C.c_void_p), ("stats_out", C.c_void_p),
        ("M", C.c_int), ("rows_per_batch", C.c_int), ("ldvt", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("a_batch_rows", C.c_int), ("a_row_off", C.c_int), ("c_batch_rows", C.c_int), ("c_row_off", C.c_int),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p), ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("residual", C.c_void_p), ("vt", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int), ("ldrb", C.c_int),
        ("rows_per_batch", C.c_int), ("flags", C.c_int),
        ("seg", C.c_int), ("period", C.c_int), ("ldvt", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("Hout", C.c_int), ("Wout", C.c_int), ("stride", C.c_int), ("up", C.c_int), ("corner_patch", C.c_int),
        ("a_batch_rows", C.c_int), ("a_row_off", C.c_int), ("c_batch_rows", C.c_int), ("c_row_off", C.c_int),
        ("gate", C.c_void_p), ("ldg", C.c_int), ("out_scale", C.c_float), ("rms_wq", C.c_void_p), ("rms_wk", C.c_void_p), ("rms_eps", C.c_float), ("vhalo", C.c_int), ("a2", C.c_void_p), ("lda2", C.c_int), ("k_split", C.c_int),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_slabs", C.c_int), ("ln_eps", C.c_float), ("stats_out", C.c_void_p),
        ("segs", C.POINTER(GemmSeg)), ("n_segs", C.c_int), ("splitk", C.c_int),
        ("ln_final", C.c_void_p), ("ln_final_out", C.c_void_p), ("ln_final_cnt", C.c_void_p), ("gn_part_out", C.c_void_p), ("cin_valid", C.c_int),
    ]


class AttnTailDesc(C.Structure):
    """mx_attn_tail_desc"""
    _fields_ = [("out1", GemmDesc), ("to_q", GemmDesc), ("out2", GemmDesc),
                ("k", C.c_void_p), ("ldk", C.c_int), ("vt", C.c_void_p), ("ldvt", C.c_int), ("vt_batch_stride", C.c_int64),
                ("B", C.c_int), ("heads", C.c_int), ("L", C.c_int), ("ctx_len", C.c_int), ("sync", C.c_void_p)]


class AttnProblem(C.Structure):
    """mx_attn_problem"""
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p), ("vt_batch_stride", C.c_int64),
                ("B", C.c_int), ("Lq", C.c_int), ("Lk", C.c_int), ("ldvt", C.c_int)]


class GnProblem(C.Structure):
    """mx_gn_problem"""
    _fields_ = [("x", C.c_void_p), ("x2", C.c_void_p), ("y", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class UNetGroup(C.Structure):
    """mx_unet_group: the samples of one resolution of a mixed batch"""
    _fields_ = [("latents", C.c_void_p), ("out", C.c_void_p), ("batch", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class UNetConfigC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("n_levels", C.c_int),
        ("block_out_channels", C.c_int * 4), ("layers_per_block", C.c_int),
        ("down_has_attn", C.c_int * 4), ("transformer_layers", C.c_int * 4), ("num_heads", C.c_int * 4),
        ("cross_attention_dim", C.c_int), ("addition_time_embed_dim", C.c_int),
        ("projection_class_embeddings_input_dim", C.c_int), ("norm_num_groups", C.c_int),
        ("norm_eps", C.c_float), ("transformer_norm_eps", C.c_float), ("layer_norm_eps", C.c_float),
    ]


class MMDiTConfigC(C.Structure):
    _fields_ = [
        ("patch_size", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int), ("num_layers", C.c_int),
        ("num_attention_heads", C.c_int), ("joint_attention_dim", C.c_int), ("pooled_projection_dim", C.c_int),
        ("pos_embed_max_size", C.c_int), ("dual_attention", C.c_int * 64), ("norm_eps", C.c_float),
    ]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class PPComm(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("all_gather", ALLGATHER_FN), ("ctx", C.c_void_p)]


ALLGATHER_INPLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
PP_SYNC, PP_WARMUP, PP_STALE = 0, 1, 2


class PPStale(C.Structure):
    _fields_ = [("state", C.c_void_p), ("state_bytes", C.c_size_t), ("mode", C.c_int), ("corrected_gn", C.c_int),
                ("all_gather_async", ALLGATHER_INPLACE_FN)]


SKIP_PREDICT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                              C.POINTER(C.c_ubyte))
MSE_UNCACHED = 9.2233720368547758e18


SKIP_OBSERVE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float))


class BlockCacheC(C.Structure):
    _fields_ = [("predict", SKIP_PREDICT_FN), ("ctx", C.c_void_p), ("state", C.c_void_p), ("state_bytes", C.c_size_t),
                ("batch_key", C.c_uint64), ("cached_key", C.c_uint64), ("cached_valid", C.c_int), ("cached_batch", C.c_int),
                ("cached_h", C.c_int), ("cached_w", C.c_int), ("blocks_run", C.c_uint), ("blocks_run_hi", C.c_uint), ("observe", SKIP_OBSERVE_FN),
                ("slots", C.POINTER(C.c_int32)), ("slot_valid", C.POINTER(C.c_ubyte)), ("n_slots", C.c_int),
                ("max_h", C.c_int), ("max_w", C.c_int), ("patches_asked", C.c_ulonglong), ("patches_total", C.c_ulonglong)]


class CLIPConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_hidden_layers", C.c_int),
                ("num_attention_heads", C.c_int), ("max_position_embeddings", C.c_int), ("hidden_act", C.c_int), ("projection_dim", C.c_int),
                ("eos_token_id", C.c_int), ("hidden_layer", C.c_int), ("layer_norm_eps", C.c_float)]


class T5ConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("d_model", C.c_int), ("d_ff", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("layer_norm_epsilon", C.c_float)]


class VAEConfigC(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("n_levels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int), ("norm_eps", C.c_float)]


class WeightEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("bytes", C.c_uint64)]


# every symbol include/mxdenoise.h declares: (name, restype, argtypes)
_vp, _i, _f, _d, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64, C.c_size_t
SYMBOLS = {
    "mx_last_error": (C.c_char_p, []),
    "mx_version": (_i, []),
    "mx_profile_enable": (_i, [_i]),
    "mx_profile_collect": (_i, [C.POINTER(C.c_double)]),
    "mx_profile_records": (_i, [C.POINTER(C.c_double), _i]),
    "mx_groupnorm_halo_workspace_bytes": (_sz, [_i, _i, _i]),
    "mx_groupnorm_halo": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "mx_halo_only": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "mx_gemm": (_i, [_vp, C.POINTER(GemmDesc)]),
    "mx_conv3x3": (_i, [_vp, C.POINTER(GemmDesc)]),
    "mx_gemm_stats_slabs": (_i, [C.POINTER(GemmDesc)]),
    "mx_gemm_ln_prefers_pass": (_i, [C.POINTER(GemmDesc)]),
    "mx_row_stats": (_i, [_vp, _vp, _i, _vp, _i, _i]),
    "mx_attention": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i, _f]),
    "mx_attention_prescaled": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i]),
    "mx_attention_cross_prescaled": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i]),
    "mx_attn_tail_sync_bytes": (_sz, [_i]),
    "mx_attn_tail_supported": (_i, [C.POINTER(AttnTailDesc)]),
    "mx_attn_tail_preferred": (_i, []),
    "mx_attn_tail": (_i, [_vp, C.POINTER(AttnTailDesc)]),
    "mx_attn_tail_status": (_i, [_vp, _vp, C.POINTER(C.c_uint)]),
    "mx_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f]),
    "mx_groupnorm_nhwc_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mx_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "mx_groupnorm_nhwc_cat": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "mx_attention_prescaled_grouped": (_i, [_vp, C.POINTER(AttnProblem), _i, _i, _i, _i, _i]),
    "mx_groupnorm_nhwc_grouped_workspace_bytes": (_sz, [C.POINTER(GnProblem), _i, _i]),
    "mx_groupnorm_nhwc_grouped": (_i, [_vp, C.POINTER(GnProblem), _i, _i, _vp, _vp, _i, _i, _f, _i, _i, _vp]),
    "mx_unet_workspace_bytes_mixed": (_sz, [_vp, C.POINTER(UNetGroup), _i, _i]),
    "mx_unet_forward_mixed": (_i, [_vp, _vp, C.POINTER(UNetGroup), _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz]),
    "mx_unet_forward_mixed_trace": (_i, [_vp, _vp, C.POINTER(UNetGroup), _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, C.c_char_p, _vp, _sz]),
    "mx_mmdit_workspace_bytes_mixed": (_sz, [_vp, C.POINTER(UNetGroup), _i, _i]),
    "mx_mmdit_forward_mixed": (_i, [_vp, _vp, C.POINTER(UNetGroup), _i, _i, _vp, _vp, _vp, _i, _vp, _sz]),
    "mx_unet_create": (_vp, [C.POINTER(UNetConfigC)]),
    "mx_unet_destroy": (None, [_vp]),
    "mx_unet_set_context_key": (_i, [_vp, C.c_uint64]),
    "mx_unet_context_stats": (_i, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "mx_unet_set_weights": (_i, [_vp, _vp, C.c_uint64, C.POINTER(WeightEntry), _i]),
    "mx_unet_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "mx_unet_validate": (_i, [_vp, _i, _i, _i, _i]),
    "mx_unet_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz]),
    "mx_forest_predict": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "mx_mmdit_workspace_bytes_pp": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "mx_mmdit_pp_state_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "mx_mmdit_forward_pp": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz]),
    "mx_mmdit_pp_comm_plan": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "mx_unet_pp_state_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "mx_unet_forward_pp_stale": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz]),
    "mx_unet_block_cache_bytes": (_sz, [_vp, _i, _i, _i]),
    "mx_gemm_splitk": (_i, [_vp, _i]),
    "mx_gemm_release_scratch": (None, [_vp, _i]),
    "mx_gemm_ln_final_supported": (_i, [C.POINTER(GemmDesc)]),
    "mx_gemm_launches": (_i, [C.POINTER(GemmDesc)]),
    "mx_gemm_gn_partials_supported": (_i, [C.POINTER(GemmDesc), _i]),
    "mx_gemm_form": (_i, [C.POINTER(GemmDesc), _i]),
    "mx_groupnorm_nhwc_from_partials": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "mx_unet_patch_cache_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "mx_mmdit_patch_cache_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "mx_mmdit_workspace_bytes_cached_mixed": (_sz, [_vp, C.POINTER(UNetGroup), _i, _i, _i]),
    "mx_mmdit_forward_cached_mixed": (_i, [_vp, _vp, C.POINTER(UNetGroup), _i, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "mx_unet_workspace_bytes_cached_mixed": (_sz, [_vp, C.POINTER(UNetGroup), _i, _i, _i]),
    "mx_unet_forward_cached_mixed": (_i, [_vp, _vp, C.POINTER(UNetGroup), _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "mx_unet_forward_cached": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mx_mmdit_block_cache_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "mx_mmdit_forward_cached": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mx_unet_forward_trace": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz,
                                   C.c_char_p, _vp, _sz]),
    "mx_unet_workspace_bytes_pp": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "mx_unet_forward_pp": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(PPComm), _vp, _sz]),
    "mx_unet_pp_comm_plan": (_i, [_vp, _i, _i, _i, _i, C.POINTER(PPComm)]),
    "mx_attention_prescaled_chunked": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64]),
    "mx_layernorm_mod": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f]),
    "mx_layernorm_mod_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _i]),
    "mx_rmsnorm_heads": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _f]),
    "mx_mmdit_create": (_vp, [C.POINTER(MMDiTConfigC)]),
    "mx_mmdit_destroy": (None, [_vp]),
    "mx_mmdit_set_weights": (_i, [_vp, _vp, C.c_uint64, C.POINTER(WeightEntry), _i]),
    "mx_mmdit_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "mx_mmdit_validate": (_i, [_vp, _i, _i, _i, _i]),
    "mx_mmdit_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz]),
    "mx_mmdit_forward_trace": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, C.c_char_p, _vp, _sz]),
    "mx_vae_create": (_vp, [C.POINTER(VAEConfigC)]),
    "mx_vae_destroy": (None, [_vp]),
    "mx_vae_set_weights": (_i, [_vp, _vp, C.c_uint64, C.POINTER(WeightEntry), _i]),
    "mx_vae_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "mx_vae_validate": (_i, [_vp, _i, _i, _i]),
    "mx_vae_decode": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sz]),
    "mx_clip_create": (_vp, [C.POINTER(CLIPConfigC)]),
    "mx_clip_destroy": (None, [_vp]),
    "mx_clip_set_weights": (_i, [_vp, _vp, C.c_uint64, C.POINTER(WeightEntry), _i]),
    "mx_clip_workspace_bytes": (_sz, [_vp, _i]),
    "mx_clip_validate": (_i, [_vp, _i]),
    "mx_clip_encode": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "mx_attention_prescaled_causal": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i]),
    "mx_attention_prescaled_bias": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i, _i, _i, _i, _i, _vp, _i]),
    "mx_rmsnorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f]),
    "mx_t5_create": (_vp, [C.POINTER(T5ConfigC)]),
    "mx_t5_destroy": (None, [_vp]),
    "mx_t5_set_weights": (_i, [_vp, _vp, C.c_uint64, C.POINTER(WeightEntry), _i]),
    "mx_t5_workspace_bytes": (_sz, [_vp, _i, _i]),
    "mx_t5_validate": (_i, [_vp, _i, _i]),
    "mx_t5_encode": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _sz]),
    "mx_cfg_flow_step": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i64, _i]),
    "mx_euler_scale_input": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i]),
    "mx_cfg_euler_step": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i64, _i]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmxdenoise.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MXDENOISE_LIB", LIB_PATH)      # override: an out-of-tree or diagnostic build of the same ABI
    if not os.path.exists(path):
        raise MxError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(or `make -C sduss_amd/csrc`). There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().mx_last_error().decode("utf-8", "replace")
        raise MxError(f"{what}: {msg}" if what else msg)


def torch_dtype_code(dtype) -> int:
    import torch
    if dtype == torch.float32:
        return MX_F32
    if dtype == torch.float16:
        return MX_F16
    if dtype == torch.bfloat16:
        return MX_BF16
    raise MxError(f"unsupported dtype {dtype}")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream

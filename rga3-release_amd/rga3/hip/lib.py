"""ctypes loader for librga3_hip.so (C ABI declared in include/rga3_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or a symbol is
absent, importing / calling raises.  Build it with ``make -C rga3-release_amd/csrc`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "librga3_hip.so"))

_p, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# symbol -> argtypes; mirrors include/rga3_hip.h one to one (tests/test_abi.py checks header <-> table <-> .so)
SIGNATURES = {
    "rga3_version": [],
    "rga3_last_error": [C.c_char_p, _sz],
    "rga3_gemm_bf16": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _p, _i64, _p],
    "rga3_gemm_rms_bf16": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _p, _i64, _p, _i64, _f, _p, _p],
    "rga3_gemm_workspace_bytes": [],
    "rga3_gemm_tn_bf16": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _p, _i64, _p, _p],
    "rga3_gemm_tn_many": [_p, _p, _i, _p, _i64, _p],
    "rga3_gemm_swiglu_pre_bf16": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i, _p, _i64, _p],
    "rga3_gemm_cat_bf16": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _i, _p],
    "rga3_attn_varlen_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64,
                             _i64, _i64, _f, _i, _i, _p, _i64, _i, _i, _i, _p],
    "rga3_attn_varlen_fwd_rope": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _p, _p, _p, _p, _p],
    "rga3_rmsnorm_fwd": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _p],
    "rga3_layernorm_fwd": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _i, _p],
    "rga3_layernorm_stats": [_p, _p, _i64, _i64, _i64, _f, _p],
    "rga3_gemm_ln_bf16": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _p],
    "rga3_gemm_lnsum_slices": [_i64, _i],
    "rga3_gemm_lnsum_bf16": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i, _p, _p],
    "rga3_gemm_lnq_bf16": [_p, _p, _p, _p, _p, _i64, _i64, _f, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _p],
    "rga3_rope_inplace": [_p, _p, _p, _i64, _i, _i, _i, _i64, _i64, _p],
    "rga3_gather_rows": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "rga3_scatter_rows": [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "rga3_pad_cols": [_p, _p, _i64, _i64, _i64, _i64, _p],
    "rga3_silu_mul": [_p, _p, _p, _i64, _p],
    "rga3_add": [_p, _p, _p, _i64, _p],
    "rga3_cross_entropy_rows": [_p, _i, _p, _p, _p, _i64, _i64, _i64, _f, _p],
    "rga3_im2col": [_p, _p, _i64, _i, _i, _i, _i, _i, _i, _i64, _p],
    "rga3_maxpool2x2_win": [_p, _p, _i64, _i, _i, _i64, _i64, _p],
    "rga3_upsample2x_add": [_p, _p, _p, _i64, _i, _i, _i, _p],
    "rga3_add_bcast": [_p, _p, _p, _i64, _i64, _i, _i64, _i64, _i64, _f, _p],
    "rga3_bilinear": [_p, _i, _p, _p, _i64, _i, _i, _i, _i, _p],
    "rga3_conv3x3s2": [_p, _i, _p, _p, _p, _i64, _i, _i, _i, _i, _f, _f, _p],
    "rga3_decimg_rows": [_p, _i64, _p, _i64, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i, _f, _i64, _p],
    "rga3_attn_fewq": [_p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _f, _p],
    "rga3_copy_many": [_p, _p, _p, _i, _p],
    "rga3_conv3x3s2_ln_gelu": [_p, _i, _p, _p, _p, _p, _f, _p, _i64, _i, _i, _i, _i, _f, _f, _p],
    "rga3_dwconv7x7": [_p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "rga3_rope_axial_inplace": [_p, _p, _p, _i64, _i, _i, _i64, _p],
    "rga3_pixel_shuffle2x": [_p, _p, _p, _p, _i64, _i, _i, _i, _i, _p],
    "rga3_bce_dice_sums": [_p, _p, _p, _i64, _i64, _p],
    "rga3_bce_dice_sums_ws_floats": [_i64, _i64],
    "rga3_bce_dice_sums_det": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "rga3_attn_varlen_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _i, _i, _p, _f, _i, _p, _i64, _p],
    "rga3_transpose16_many": [_p, _p, _i, _p],
    "rga3_hiera_mlp144": [_p, _p, _p, _p, _p, _p, _p, _i64, _f, _p],
    "rga3_hiera_mlp288_pack_bytes": [],
    "rga3_hiera_mlp288_pack": [_p, _p, _p, _p, _p, _p],
    "rga3_hiera_mlp288": [_p, _p, _p, _p, _i64, _f, _p],
    "rga3_mlp3_rows": [_p, _p, _i, _i64, _p],
    "rga3_sam_select_objptr": [_p, _p, _p, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p],
    "rga3_memattn_cross_ws_floats": [_i64, _i],
    "rga3_memattn_cross": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _p, _p],
    "rga3_memlayer_rows": [_p, _i64, _i, _p, _p, _i, _p, _p, _p, _i64, _p, _i64, _p, _p, _f, _p, _i64, _p, _p, _i, _p, _i64, _p, _p, _i, _i, _i64, _p],
    "rga3_gemm_rows16_many": [_p, _p, _i, _p],
    "rga3_gemm_stream_k_timeouts": [_p],
    "rga3_gemm_timeout_counter_offset": [],
    "rga3_gemm_ragged_plan": [_i64, _i64, _i64, _i, _i, _p, _p],
    "rga3_quant_fp8_rows": [_p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "rga3_gemm_fp8": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "rga3_swiglu_fwd_quant_fp8": [_p, _p, _p, _i64, _i64, _p],
    "rga3_swiglu_bwd_quant_fp8": [_p, _p, _p, _p, _i64, _i64, _p],
    "rga3_im2col3x3s2": [_p, _p, _i64, _i, _i, _i, _p],
    "rga3_pil_bicubic_coeffs": [_i, _i, _p, _p, _i64, _p],
    "rga3_sam_preprocess_u8": [_p, _i64, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p],
    "rga3_qwen_norm_lut": [_p, _p, _i, _p],
    "rga3_qwen_patchify_u8": [_p, _i64, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "rga3_rmsnorm_bwd": [_p, _p, _p, _p, _p, _i64, _i64, _f, _p],
    "rga3_dropout_bf16": [_p, _p, _i64, _f, _i64, _i, _p],
    "rga3_dropout_pair_bf16": [_p, _p, _p, _p, _i64, _f, _i64, _f, _i64, _i, _p],
    "rga3_swiglu_fwd": [_p, _p, _i64, _i64, _p],
    "rga3_swiglu_bwd": [_p, _p, _p, _i64, _i64, _p],
    "rga3_transpose16": [_p, _p, _i64, _i64, _i64, _i64, _p],
    "rga3_segment_sum_rows": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "rga3_adamw_step": [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i, _f, _p],
    "rga3_sumsq_accum": [_p, _p, _i64, _p],
    "rga3_sumsq_det": [_p, _i64, _p, _i64, _p, _i, _p],
    "rga3_adamw_step_clip": [_p, _p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i, _p, _f, _p],
    "rga3_adamw_step_clip_rows": [_p, _p, _p, _p, _p, _i64, _i64, _p, _f, _f, _f, _f, _i, _p, _f, _p],
    "rga3_scatter_add_rows": [_p, _p, _p, _i64, _i64, _i64, _i64, _f, _p],
    "rga3_layernorm_bwd_ws_floats": [_i64, _i64],
    "rga3_layernorm_bwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _f, _p, _i64, _p],
    "rga3_colsum_accum": [_p, _p, _i64, _i64, _i64, _p],
    "rga3_colsum_ws_floats": [_i64, _i64],
    "rga3_colsum": [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _p],
    "rga3_act": [_p, _p, _p, _i64, _i, _p],
    "rga3_bilinear_bwd": [_p, _p, _p, _i64, _i, _i, _i, _i, _p],
    "rga3_mask_product": [_p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "rga3_mask_product_bwd_ws_floats": [_i64, _i64, _i64, _i64],
    "rga3_mask_product_bwd": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p],
    "rga3_pixel_shuffle2x_bwd": [_p, _p, _i64, _i, _i, _i, _p],
    "rga3_bce_dice_grad": [_p, _p, _p, _p, _i64, _i64, _f, _f, _p],
    "rga3_bce_dice_grad_dev": [_p, _p, _p, _p, _i64, _i64, _p, _p, _p],
}

_INT64_RESULTS = ("rga3_gemm_lnsum_slices", "rga3_hiera_mlp288_pack_bytes", "rga3_gemm_workspace_bytes", "rga3_memattn_cross_ws_floats", "rga3_gemm_timeout_counter_offset", "rga3_layernorm_bwd_ws_floats", "rga3_colsum_ws_floats", "rga3_mask_product_bwd_ws_floats", "rga3_bce_dice_sums_ws_floats")

_lib = None


class Rga3Error(RuntimeError):
    pass


def load():
    """Load the library once; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Rga3Error(
            f"{LIB_PATH} not found: the HIP extension is required (no fallback path). "
            "Run `make -C rga3-release_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    # The library binds to whichever HIP runtime (SONAME libamdhip64.so.7) the process loaded first.  PyTorch ships
    # its own copy; load torch's first so kernels, streams and events all live in ONE runtime (loading ours first
    # would pull /opt/rocm's copy in and split the process across two runtimes).
    import torch  # noqa: F401

    _thip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(_thip):
        C.CDLL(_thip, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in _INT64_RESULTS else C.c_int
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().rga3_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise Rga3Error(f"{what} failed (rc={rc}): {last_error()}")

"""ctypes binding of libvgpt_hip.so (C ABI declared in include/vgpt.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  `load()` only dlopens the library (no GPU needed); compute entry points
need a gfx950 device.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGPT_LIB selects another build of the same library (the `make stamps` diagnostic build); never a different backend
LIB_PATH = os.environ.get("VGPT_LIB") or os.path.join(_HERE, "libvgpt_hip.so")

# enums mirrored from include/vgpt.h
ACT_SILU, ACT_GELU, ACT_GELU_TANH, ACT_NONE = 0, 1, 2, 3
EPI_NONE, EPI_RESID, EPI_BIAS = 0, 1, 2
PRED_V, PRED_X1 = 0, 1

_P = c_void_p
_I64 = c_int64
# VGPT_ABI_VERSION (include/vgpt.h) the SIGNATURES table below was written for; load() refuses any other library
ABI_VERSION = 5

# name -> (restype, argtypes); every symbol declared in include/vgpt.h
SIGNATURES = {
    "vgpt_last_error": (c_char_p, []),
    "vgpt_abi_version": (c_int, []),
    "vgpt_rmsnorm_fwd": (c_int, [_P, _P, _P, _I64, _I64, c_float, _P]),
    "vgpt_rope_table": (c_int, [_P, _P, _P, _P, _I64, c_int, c_int, c_float, _P]),
    "vgpt_rope_qk_inplace": (c_int, [_P, _P, _P, _I64, c_int, c_int, c_int, _P]),
    "vgpt_gemm_bf16": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P]),
    "vgpt_gemm_bf16_rope": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_int, _P]),
    "vgpt_gemm_bf16_tr": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_int, c_int, _P]),
    "vgpt_gemm_set_family": (c_int, [c_int]),
    "vgpt_gemm_norm_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "vgpt_gemm_bf16_resid_rstd": (c_int, [_P, _P, _P, _P, _P, _P, _I64, c_float, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P]),
    "vgpt_rms_rstd": (c_int, [_P, _P, _I64, _I64, _I64, c_float, _P]),
    "vgpt_fold_norm_gain": (c_int, [_P, _P, _P, _I64, _I64, _P]),
    "vgpt_gemm_bf16_rope_prenorm": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, c_int, _P]),
    "vgpt_gated_mlp_act_fwd_prenorm": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P]),
    "vgpt_gated_mlp_act_fwd": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P]),
    "vgpt_gated_mlp_act_fwd_keep": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, c_int, _P]),
    "vgpt_mask_pack_bool": (c_int, [_P, _P, _I64, _I64, _P]),
    "vgpt_mask_pack_additive": (c_int, [_P, c_int, _P, _I64, _I64, _P]),
    "vgpt_mask_build_tokens": (c_int, [_P, _P, _I64, _I64, _P]),
    "vgpt_mask_tile_summary": (c_int, [_P, _P, _I64, _I64, _P]),
    "vgpt_mask_count_empty_rows": (c_int, [_P, _P, _I64, _I64, _P]),
    "vgpt_attn_blockmask_fwd": (
        c_int,
        [_P, _P, _P, _P, _P, _P, _I64, _I64, c_int, c_int, c_int] + [_I64] * 12 + [c_float, c_int, _P],
    ),
    "vgpt_attn_supported": (c_int, [c_int]),
    "vgpt_attn_blockmask_fwd_qrange": (
        c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I64, c_int, c_int, c_int] + [_I64] * 12 + [c_float, _P]),
    "vgpt_attn_qblock_order": (c_int, [_P, _I64, _I64, _I64, _P, _P]),
    "vgpt_attn_trace": (c_int, [_P, _I64]),
    "vgpt_attn_set_hand_scheduled": (c_int, [c_int]),
    "vgpt_attn_plan_build": (c_int, [_P, _I64, _I64, _P, _I64, _P, _P, _P]),
    "vgpt_attn_plan_workspace_bytes": (_I64, [_I64, _I64]),
    "vgpt_attn_fwd_plan": (
        c_int, [_P] * 9 + [_I64, _I64, _I64, c_int, c_int, c_int] + [_I64] * 12 + [c_float, c_int, _P]),
    "vgpt_attn_fp8_workspace_bytes": (_I64, [_I64, _I64, c_int, c_int, c_int]),
    "vgpt_attn_fp8_quantize": (c_int, [_P] * 4 + [_I64, _I64, _I64, c_int, c_int, c_int] + [_I64] * 9 + [c_float, _P]),
    "vgpt_attn_fwd_plan_fp8": (c_int, [_P] * 6 + [_I64, _I64, _I64, c_int, c_int, c_int] + [_I64] * 3 + [_P]),
    "vgpt_embed_gather": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "vgpt_patch_embed_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64, c_int, _P]),
    "vgpt_timestep_sinusoid": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "vgpt_linear_small": (c_int, [_P, _P, _P, _P, _P, c_int, _I64, _I64, _I64, _I64, c_int, c_int, _P]),
    "vgpt_final_layer_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64, c_float, _P]),
    "vgpt_sampler_set_timesteps": (c_int, [_P, _P, c_int, _P, c_int, _P]),
    "vgpt_euler_cfg_update": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _I64, c_int, c_int, c_float, _P]),
    "vgpt_sampler_advance": (c_int, [_P, _P]),
    "vgpt_sampler_copy_step_rows": (c_int, [_P, _P, _P, c_int, c_int, _I64, _I64, _I64, _I64, _P]),
    "vgpt_cast_f32_to_bf16": (c_int, [_P, _P, _I64, _P]),
    "vgpt_groupnorm_stats": (c_int, [_P, _P, _I64, c_int, c_int, c_int, c_float, _P]),
    "vgpt_conv2d_fwd": (c_int, [_P] * 8 + [c_int] * 11 + [_I64, _I64, _P]),
    "vgpt_conv_bx3_packed_bytes": (c_int64, [c_int, c_int]),
    "vgpt_conv_pack_weights_bx3": (c_int, [_P, _P, c_int, c_int, _P]),
    "vgpt_conv2d_bx3_fwd": (c_int, [_P] * 8 + [c_int] * 8 + [_P]),
    "vgpt_conv1x1_bx3_packed_bytes": (c_int64, [c_int, c_int]),
    "vgpt_conv1x1_pack_weights_bx3": (c_int, [_P, _P, c_int, c_int, _P]),
    "vgpt_conv1x1_bx3_fwd": (c_int, [_P] * 8 + [c_int] * 6 + [_P]),
    "vgpt_col_softmax": (c_int, [_P, c_int, c_int, c_int, c_float, _P]),
    "vgpt_vae_sample": (c_int, [_P, _P, _P, c_int, _I64, c_float, c_float, _P]),
    "vgpt_vae_postprocess_u8": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vgpt_affine_to_f32": (c_int, [_P, c_int, _P, _I64, c_float, c_float, _P]),
    "vgpt_attn_blockmask_fwd_lse": (
        c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, c_int, c_int, c_int] + [_I64] * 12 + [c_float, _P]),
    "vgpt_attn_blockmask_bwd": (
        c_int, [_P] * 12 + [_I64, _I64, c_int, c_int, c_int, _P, c_float, _P]),
    "vgpt_transpose_pad_bf16": (c_int, [_P, _P, _I64, _I64, _I64, _I64, _P]),
    "vgpt_silu_mul_fwd": (c_int, [_P, _P, _I64, _I64, c_int, _P]),
    "vgpt_silu_mul_bwd": (c_int, [_P, _P, _P, _I64, _I64, c_int, _P]),
    "vgpt_act_fwd": (c_int, [_P, _P, _I64, c_int, _P]),
    "vgpt_act_bwd": (c_int, [_P, _P, _P, _I64, c_int, _P]),
    "vgpt_rmsnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, c_float, _P]),
    "vgpt_matmul_generic_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "vgpt_matmul_generic": (c_int, [_P, c_int, _I64, _I64, _P, c_int, _I64, _I64, _P, c_int, _I64, _I64, _I64, _I64,
                                    _I64, c_float, c_int, _P, _I64, _P]),
    "vgpt_colsum": (c_int, [_P, c_int, _P, _I64, _I64, _I64, c_int, _P]),
    "vgpt_lerp_frames": (c_int, [_P, _P, _P, _P, c_int, _I64, _P]),
    "vgpt_mse_frames": (c_int, [_P, _P, _P, _P, c_int, _I64, _P]),
    "vgpt_mse_frames_mean": (c_int, [_P, _P, _P, _P, c_int, c_int, _I64, _P]),
    "vgpt_ln_mod_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _I64, c_float, _P]),
    "vgpt_ln_mod_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I64, _P]),
    "vgpt_embed_bwd": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "vgpt_patchify": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vgpt_unpatchify_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "vgpt_gather_rows": (c_int, [_P, _P, _P, c_int, c_int, _I64, _P]),
    "vgpt_sumsq": (c_int, [_P, c_int, _P, _I64, _P, _P]),
    "vgpt_clip_coef": (c_int, [_P, _P, _P, c_float, c_float, _P]),
    "vgpt_adamw_step": (c_int, [_P, _P, _P, c_int, _P, _P, _I64, c_float, c_float, c_float, c_float, c_float, c_int,
                                _P, _P]),
    "vgpt_graph_begin_capture": (c_int, [_P]),
    "vgpt_graph_end_capture": (c_int, [_P, POINTER(c_void_p)]),
    "vgpt_graph_launch": (c_int, [_P, _P]),
    "vgpt_graph_destroy": (c_int, [_P]),
    "vgpt_calib_mfma": (c_int, [_P, c_int, _P]),
    "vgpt_calib_mfma_flops": (c_double, [c_int]),
    "vgpt_calib_copy": (c_int, [_P, _P, _I64, _P]),
}


class VgptError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """dlopen libvgpt_hip.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VgptError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch first: it brings its own copy of the HIP runtime, and the one that is loaded first serves the whole process;
    # with ours (/opt/rocm) loaded first, torch's device state and our launches end up in different runtimes
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    ver = getattr(lib, "vgpt_abi_version", None)
    got = None
    if ver is not None:
        ver.restype, ver.argtypes = c_int, []
        got = ver()
    if got != ABI_VERSION:
        raise VgptError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION}: the library is "
                        "stale (rebuild it: `python -c 'import __graft_entry__ as g; g.build()'`); calling it with shifted "
                        "arguments would fault on the device")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # reported by check_exports(); calling it raises AttributeError
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def missing_exports():
    lib = load()
    return [n for n in SIGNATURES if not hasattr(lib, n)]


def call(name: str, *args) -> None:
    """Call an int-returning entry point, raising VgptError with vgpt_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.vgpt_last_error()
        raise VgptError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")

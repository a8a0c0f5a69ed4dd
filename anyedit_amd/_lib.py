"""ctypes binding of libanyedit_hip.so (include/anyedit_hip.h).

There is NO fallback: if the HIP extension is missing or fails to load, importing this module raises.
`import torch` happens first on purpose: the extension must bind to the same libamdhip64 the torch
allocator/streams live in (one HIP runtime per process).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AE_LIB_PATH") or os.path.join(_HERE, "libanyedit_hip.so")  # AE_LIB_PATH: dev knob (A/B of two builds on one box)

c_void_p, c_int, c_long, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/anyedit_hip.h 1:1
SIGNATURES = {
    "ae_version": [],
    "ae_last_error": [],
    "ae_device_arch": [ctypes.c_char_p, c_int],
    "ae_device_info": [ctypes.POINTER(c_int), ctypes.POINTER(c_long), ctypes.POINTER(c_int)],
    "ae_gemm_bf16": [c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int,
                     c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p],
    "ae_conv3x3_workspace_floats": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ae_conv3x3_partials_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    "ae_groupnorm_splitk_supported": [c_int, c_int, c_int, c_int, c_int],
    "ae_groupnorm_splitk_nhwc_bf16": [c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "ae_conv3x3_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                        c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "ae_conv3x3_up2_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ae_groupnorm_rows_per_chunk": [c_int, c_int],
    "ae_groupnorm_workspace_floats": [c_int, c_int, c_int, c_int],
    "ae_groupnorm_nhwc_bf16": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ae_layernorm_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "ae_attn_fwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                         c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long, c_long,
                         c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                         c_void_p, c_void_p, c_int, c_long, c_long, c_long, c_long, c_long, c_long, c_void_p, c_void_p, c_void_p,
                         c_void_p],
    "ae_attn_bwd_bf16": [c_void_p] * 10 + [c_int] * 5 + [c_long] * 21 + [c_float, c_void_p, c_int, c_void_p, c_void_p],
    "ae_attn_bwd_workspace_floats": [c_int] * 5,
    "ae_groupnorm_bwd_workspace_floats": [c_int, c_int, c_int, c_int],
    "ae_groupnorm_bwd_nhwc_bf16": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "ae_layernorm_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p],
    "ae_layernorm_param_grad_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "ae_add_bf16": [c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ae_axpy_bf16": [c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p],
    "ae_geglu_fwd_bf16": [c_void_p, c_void_p, c_long, c_int, c_void_p],
    "ae_geglu_bwd_bf16": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "ae_sumpool2x2_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "ae_colsum_bf16_f32": [c_void_p, c_void_p, c_int, c_int, c_long, c_void_p],
    "ae_mse_grad_f32": [c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p],
    "ae_adamw_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float, c_int, c_float,
                     c_void_p],
    "ae_rowsum_f32": [c_void_p, c_void_p, c_int, c_long, c_void_p],
    "ae_scatter_add_rows_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "ae_attn_fp8_workspace_bytes": [c_int, c_int, c_int, c_int, c_int],
    "ae_attn_fwd_fp8": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_long] * 12 +
                       [c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_long, c_void_p],
    "ae_ln_gemm_supported": [c_int, c_int, c_int, c_int],
    "ae_ln_gemm_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_long, c_void_p, c_void_p,
                        c_float, c_int, c_void_p, c_void_p],
    "ae_gemm_ln_plan": [c_int, c_int, c_int, c_int, c_int],
    "ae_xattn_fused_supported": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ae_xattn_fused_covers": [c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "ae_xattn_fused_kv_bytes": [],
    "ae_xattn_fused_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int,
                            c_float, c_void_p],
    "ae_ff_fused_supported": [c_int, c_int, c_int],
    "ae_ff_fused_bf16": [c_void_p, c_long, c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long,
                         c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p],
    "ae_gemm_ln_bf16": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_long, c_int, c_void_p,
                        c_void_p, c_int, c_void_p, c_float, c_void_p],
    "ae_task_gate_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ae_task_gate_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ae_expert_kv_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_expert_kv_dgrad_slices": [c_int],
    "ae_expert_kv_dgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ae_expert_kv_wgrad": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_transpose_last2": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_concat_channels_bf16": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_void_p],
    "ae_im2col3x3_c8_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "ae_split_channels_bf16": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "ae_timestep_embedding": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "ae_ddim_step_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_float,
                         c_float, c_float, c_float, c_float, c_float, c_void_p],
    "ae_ddim_encode_step_f32": [c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_float, c_float, c_void_p],
    "ae_plms_combine_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p],
    "ae_mask_blend_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p],
    "ae_q_sample_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p],
    "ae_silu_to_bf16": [c_void_p, c_int, c_void_p, c_long, c_void_p],
    "ae_add_bcast_bf16": [c_void_p, c_void_p, c_void_p, c_long, c_long, c_void_p],
    "ae_resample2x_rows_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_scale_shift_rows_bf16": [c_void_p, c_void_p, c_long, c_void_p, c_int, c_long, c_int, c_int, c_void_p],
    "ae_window_partition_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_layernorm_window_supported": [c_int],
    "ae_layernorm_window_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_void_p],
    "ae_sam_relpos_terms": [c_void_p, c_long, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_void_p],
    "ae_softmax_rows_f32_bf16": [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_float, c_void_p],
    "ae_gaussian_moments_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p],
    "ae_ms_deform_attn_fwd_f32": [c_void_p] * 6 + [c_int] * 7 + [c_void_p],
    "ae_ms_deform_attn_bwd_f32": [c_void_p] * 9 + [c_int] * 7 + [c_void_p],
    "ae_linear_f32": [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p],
    "ae_dpm_multistep_f32": [c_void_p] * 5 + [c_long, c_int, c_float, c_int, c_int, c_float, c_float, c_int, c_float, c_float, c_float,
                             c_float, c_void_p],
    "ae_layernorm_act_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float, c_int, c_void_p],
    "ae_sam_pe_encode_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p],
    "ae_sam_mask_downscale_bf16": [c_void_p] * 10 + [c_int, c_int, c_int, c_float, c_void_p],
    "ae_sam_mask_product_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_sam_postprocess_masks": [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_float, c_int, c_void_p],
    "ae_nms_sorted_f32": [c_void_p, c_void_p, c_int, c_float, c_void_p],
    "ae_sam_preprocess_f32": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ae_patchify_f32_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "ae_mse_f32": [c_void_p, c_void_p, c_void_p, c_long, c_void_p],
    "ae_lincomb4_f32": [c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p, c_float, c_long, c_void_p],
    "ae_dpm_adaptive_err_f32": [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_long, c_void_p, c_void_p],
    "ae_task_gate": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
}
_RESTYPES = {"ae_last_error": ctypes.c_char_p, "ae_groupnorm_workspace_floats": c_long, "ae_conv3x3_workspace_floats": c_long,
             "ae_groupnorm_bwd_workspace_floats": c_long, "ae_attn_fp8_workspace_bytes": c_long, "ae_attn_bwd_workspace_floats": c_long,
             "ae_xattn_fused_kv_bytes": c_long}


class AnyEditHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is REQUIRED (no CPU/eager fallback exists). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `python anyedit_amd/build.py`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.ae_last_error()
        raise AnyEditHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def device_arch():
    buf = ctypes.create_string_buffer(128)
    check(lib.ae_device_arch(buf, 128), "ae_device_arch")
    return buf.value.decode()

"""ctypes binding of libdehaze_hip.so (C-ABI declared in include/dehaze_hip.h).

There is NO fallback: if the shared object is missing the import of the product fails loudly with
the build instruction; nothing here routes to PyTorch eager or to the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DHZ_LIB_PATH") or os.path.join(_HERE, "libdehaze_hip.so")   # override: A/B of two builds

c_f = ctypes.c_void_p      # device float*
c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_fl = ctypes.c_float

# name -> argtypes  (return type is int unless listed in _RESTYPE)
SIGNATURES = {
    "dhz_abi_version": [],
    "dhz_last_error": [],
    "dhz_build_id": [],
    "dhz_set_reserved_cus": [ctypes.c_int],
    "dhz_gelu_fwd_dt": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p],
    "dhz_gelu_bwd_dt": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p],
    "dhz_get_reserved_cus": [],
    "dhz_grid_cus": [],
    "dhz_ps_attn_fwd": [c_f, c_f, c_f, c_i, c_p, c_f, c_f, c_f, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    "dhz_ps_attn_bwd_parts": [c_i, c_i],
    "dhz_ps_attn_bwd_parts_d": [c_i, c_i, c_i],
    "dhz_ps_attn_bwd": [c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_fused_attn_prepack": [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_p],
    "dhz_fused_window_attn_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p,
                                  c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_fused_window_attn_fwd6": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p,
                                   c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_fused_attn_prepack6": [c_f, c_f, c_f, c_f, c_p, c_i, c_p],
    "dhz_fused_attn_prepack6_multi": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "dhz_fused_attn_bwd_parts": [c_i],
    "dhz_fused_attn_bwd_prepack": [c_f, c_f, c_f, c_f, c_f, c_i, c_p],
    "dhz_fused_window_attn_bwd": [c_f] * 11 + [c_f] * 12 + [c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_dense_attn_fwd": [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_p],
    "dhz_dense_attn_bwd": [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_fl, c_p],
    "dhz_bias_gather": [c_f, c_f, c_i, c_p],
    "dhz_bias_gather_multi": [c_p, c_p, c_p, c_i, c_p],
    "dhz_fused_attn_prepack_multi": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "dhz_bias_table_grad": [c_f, c_i, c_f, c_i, c_i, c_p],
    "dhz_bias_table_grad_multi": [c_p, c_p, c_p, c_p, c_i, c_p],
    "dhz_shift_mask": [c_f, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_dgrad": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_conv3x3_in3_blocked": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_dgrad_blocked": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_wgrad": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_l1_pair_fwd": [c_f, c_f, c_f, c_f, c_l, c_p],
    "dhz_l1_pair_bwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_p],
    "dhz_contrast_combine_fwd": [c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_p],
    "dhz_contrast_combine_bwd": [c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_p],
    "dhz_crop_augment_pair": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_winograd_prepack": [c_f, c_f, c_i, c_i, c_i, c_p],
    "dhz_winograd_conv3x3": [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_winograd43_prepack": [c_f, c_f, c_i, c_i, c_i, c_p],
    "dhz_winograd43_conv3x3": [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_winograd43_conv3x3_pool": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_maxpool2x2_blocked_fwd": [c_f, c_f, c_i, c_i, c_i, c_p],
    "dhz_maxpool2x2_blocked_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_p],
    "dhz_layout_blocked8": [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_p],
    "dhz_leff_fused_fwd": [c_f] * 16 + [c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_fused_fwd6": [c_f] * 15 + [c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_prepack6": [c_f, c_f, c_p, c_i, c_p],
    "dhz_linear_fwd": [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_dgrad": [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_input_proj_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_p],
    "dhz_input_proj_bwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_p],
    "dhz_conv4s2_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_conv4s2_dgrad": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_conv4s2_wgrad": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_input_proj_fwd_dt": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_p],
    "dhz_input_proj_bwd_dt": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_p],
    "dhz_thin_conv3x3_fwd_dt": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_dgrad_dt": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_thin_conv3x3_wgrad_dt": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_vgg_prepack_bf16": [c_f, c_f, c_i, c_i, c_i, c_p],
    "dhz_vgg_conv3x3_bf16": [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_maxpool2x2_nhwc_bf16_fwd": [c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_maxpool2x2_nhwc_bf16_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_l1_pair_fwd_bf16": [c_f, c_f, c_f, c_f, c_l, c_p],
    "dhz_l1_pair_bwd_bf16": [c_f, c_f, c_f, c_f, c_f, c_l, c_p],
    "dhz_im2col_k4s2_bf16": [c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_col2im_k4s2_bf16": [c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_fwd_split": [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_dgrad_split": [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_wgrad_split": [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_p],
    "dhz_linear_fwd_split6": [c_f, c_i, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_dgrad_split6": [c_f, c_i, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_fwd_split6_res": [c_f, c_i, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_fwd_split_res": [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_fwd_bf16_res": [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_dgrad_split_scaled": [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_ln_partition_bwd_lay2": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_p],
    "dhz_ln_partition_bwd_lay": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_split3_planes": [c_f, c_l, c_p, c_p, c_p, c_p],
    "dhz_split3_planes_t": [c_f, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "dhz_linear_fwd_bf16": [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_bf16_transpose_batched": [c_f, c_p, c_p, c_i, c_i, c_p],
    "dhz_linear_dgrad_bf16": [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_wgrad_bf16": [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "dhz_ln_partition_fwd_dt": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_ln_partition_bwd_dt": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_reverse_residual_fwd_dt": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_reverse_residual_bwd_dt": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_dwconv_fwd_dt": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_dwconv_bwd_dt": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_dwconv_bwd_scaled_dt": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_ps_attn_fwd_dt": [c_f, c_f, c_f, c_i, c_p, c_f, c_f, c_f, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_ps_attn_bwd_dt": [c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_linear_wgrad": [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_p],
    "dhz_linear_wgrad_rs": [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_p],
    "dhz_linear_wgrad_multi": [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "dhz_ln_partition_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_ln_partition_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_reverse_residual_fwd": [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_reverse_residual_bwd": [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_dwconv_fwd": [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_leff_dwconv_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p],
    "dhz_charbonnier_fwd": [c_f, c_f, c_f, c_f, c_l, c_fl, c_i, c_p],
    "dhz_charbonnier_bwd": [c_f, c_f, c_f, c_f, c_f, c_l, c_fl, c_fl, c_i, c_p],
    "dhz_adamw_step": [c_f, c_f, c_f, c_f, c_l, c_fl, c_fl, c_fl, c_fl, c_fl, c_i, c_fl, c_p],
    "dhz_comm_unique_id": [c_p],
    "dhz_comm_init": [c_p, c_i, c_i, c_p],
    "dhz_comm_allreduce_sum_f32": [c_p, c_f, c_l, c_p],
    "dhz_comm_destroy": [c_p],
    "dhz_adamw_step_shadow": [c_f, c_f, c_f, c_f, c_f, c_l, c_fl, c_fl, c_fl, c_fl, c_fl, c_i, c_fl, c_p],
}
_RESTYPE = {"dhz_last_error": ctypes.c_char_p, "dhz_build_id": ctypes.c_char_p}

_lib = None


class DehazeHipError(RuntimeError):
    pass


def load():
    """Load the shared object (once).  Raises ImportError with the build recipe if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (there is no PyTorch/CPU fallback). "
            "Build it with `python __graft_entry__.py build` or `csrc/build.sh` (hipcc --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dhz_last_error()
        raise DehazeHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)

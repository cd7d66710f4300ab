"""ctypes binding of libim2im_uq.so (the C ABI in include/im2im_uq.h).

There is no CPU fallback: if the library is missing, import of the product path fails loudly.
PyTorch is used only as plumbing here (device memory, the current HIP stream).
"""
from __future__ import annotations

import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IM2IM_LIB") or os.path.join(_PKG, "lib", "libim2im_uq.so")   # IM2IM_LIB: A/B builds (tools/)


class Im2ImError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. Run `python -m im2im_uq_amd.build` "
            "(or __graft_entry__.build()). There is deliberately no CPU/PyTorch fallback.")
    return ctypes.CDLL(LIB_PATH)


lib = _load()

_i32, _i64, _f32, _f64, _ptr = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/im2im_uq.h declares (tests/test_abi.py)
SIGNATURES = {
    "im2im_abi_version": (_i32, []),
    "im2im_last_error": (ctypes.c_char_p, []),
    "im2im_rcps_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "im2im_rcps_loss_table": (_i32, [_ptr, _ptr, _i64, _i64, _ptr, _i32, _i32, _ptr, _ptr, _ptr, _ptr]),
    "im2im_rcps_miscoverage": (_i32, [_ptr, _ptr, _i64, _i32, _i64, _f32, _i32, _ptr, _ptr]),
    "im2im_nested_sets": (_i32, [_ptr, _i64, _i64, _f32, _i32, _ptr, _ptr, _i32, _i32, _ptr]),
    "im2im_fraction_missed": (_i32, [_ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr]),
    "im2im_hb_mu_plus": (_f64, [_f64, _i64, _f64, _i32]),
    "im2im_hb_mu_plus_batch": (_i32, [_ptr, _i64, _i64, _f64, _i32, _ptr]),
    "im2im_set_option": (_i32, [ctypes.c_char_p, _i32]),
    "im2im_rcps_scan": (_i32, [_ptr, _i64, _i32, _i64, _i64, _ptr, _f64, _f64, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "im2im_pack_conv_weight": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    "im2im_pack_conv_weights_multi": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr]),
    "im2im_conv_stats_rows": (_i64, [_i32, _i32, _i32, _i32]),
    "im2im_conv_fwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    # x, in_ss, x_hi, in_ss_hi, Ci_lo, wf, bias, center, scale, shift, y, y_hi, Co_lo, stats, B, H, W, Ci, Co, taps, relu, dtype, stream
    "im2im_conv_fwd_split": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr,
                                    _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_conv_fwd_eval_pool": (_i32, [_ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_conv_fwd_eval_tail": (_i32, [_ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    # x, x_ss, x_hi, x_ss_hi, Ci_lo, dz, amax_prev, dw, ws, ws_bytes, B, H, W, Ci, Co, target_wgs, nsplit (host int*), stream
    "im2im_conv_wgrad_fp8": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _ptr, _ptr]),
    "im2im_conv_splitk_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "im2im_conv_fwd_split_ws": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr,
                                       _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr, _i64, _ptr]),
    "im2im_conv_fp8_stats_rows": (_i64, [_i32, _i32, _i32]),
    "im2im_pack_conv_weight_fp8": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    # x, in_ss, x_hi, in_ss_hi, Ci_lo, wq, wscale, bias, scale, shift, y, stats, B, H, W, Ci, Co, relu, stream
    "im2im_pack_conv_weight_fp8_dgrad": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    "im2im_pack_conv_weights_fp8_multi": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    # dz, wq_d, wscale_d, dx, dx_hi, Cx_lo, amax_prev, amax_now, amax_next, B, H, W, Cz, Cx, stream
    "im2im_conv_dgrad_fp8": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_conv_fwd_fp8": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32,
                                  _i32, _ptr]),
    # dz, wd, dx, bn_z, bn_ss, bn_mi, bn_partial, B, H, W, Ci, Co, taps, dtype, stream
    "im2im_conv_dgrad_bn": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_bn_relu_bwd_from_partial": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _i64, _ptr, _ptr]),
    "im2im_conv_wgrad_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    # x, x_ss, x_hi, x_ss_hi, Ci_lo, dz, dw, ws, ws_bytes, B, H, W, Ci, Co, taps, dtype, target_wgs, nsplit (host int*), stream
    "im2im_conv_wgrad_split": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                      _i32, _i32, _ptr, _ptr]),
    "im2im_wgrad_reduce_multi": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "im2im_conv_wgrad": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_reduce_workspace_bytes": (_i64, [_i64]),
    # ..., mean_invstd, scale_shift, ws, counters, num_batches_tracked, stream
    "im2im_bn_finalize": (_i32, [_ptr, _i64, _i32, _i64, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "im2im_bn_fold_eval": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _f32, _i32, _ptr, _ptr]),
    "im2im_bn_relu_apply": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr]),
    "im2im_bn_bwd_workspace_bytes": (_i64, [_i64, _i32]),
    "im2im_bn_relu_pool_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "im2im_bn_relu_pool_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr, _i64, _ptr, _ptr]),
    "im2im_bn_relu_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _i64, _ptr, _ptr]),
    "im2im_bn_bwd_rows_per_block": (_i64, [_i64]),
    "im2im_bn_relu_bwd_phase": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _i32, _ptr, _i64, _i32, _i64, _i64, _ptr, _ptr]),
    "im2im_conv_tiles_per_image": (_i64, [_i32, _i32]),
    "im2im_conv_fwd_per_image": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_groupnorm_stats_rows": (_i64, [_i32, _i64]),
    "im2im_groupnorm_stats": (_i32, [_ptr, _i32, _i64, _i32, _i32, _ptr, _ptr]),
    "im2im_groupnorm_finalize": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _f32, _ptr, _ptr, _ptr]),
    "im2im_affine_relu_apply_per_image": (_i32, [_ptr, _ptr, _ptr, _i32, _i64, _i32, _i32, _ptr]),
    "im2im_groupnorm_relu_bwd_workspace_bytes": (_i64, [_i32, _i64, _i32]),
    "im2im_groupnorm_relu_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i64, _i32, _i32, _i32, _ptr, _i64, _ptr]),
    "im2im_head_activation_fwd": (_i32, [_ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr]),
    "im2im_head_activation_bwd": (_i32, [_ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr]),
    "im2im_depth_space2": (_i32, [_ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_colsum_workspace_bytes": (_i64, [_i64, _i32]),
    "im2im_colsum": (_i32, [_ptr, _ptr, _i64, _i32, _i32, _ptr, _i64, _ptr]),
    "im2im_maxpool2_fwd": (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_maxpool2_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_upsample2x_concat_fwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_upsample2x_concat_bwd": (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_smallconv_tiles": (_i64, [_i32, _i32, _i32]),
    "im2im_smallconv_s2l_fwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_smallconv_l2s_fwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_smallconv_wgrad_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "im2im_smallconv_wgrad": (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr, _i64, _ptr]),
    "im2im_quantile_loss_workspace_bytes": (_i64, []),
    "im2im_quantile_loss_fwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _ptr, _ptr]),
    "im2im_quantile_loss_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "im2im_softmax_ce_fwd": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    "im2im_softmax_ce_bwd": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    "im2im_softmax_sets_summary": (_i32, [_ptr, _i64, _i64, _i32, _i32, _i32, _ptr, _ptr]),
    # kind, a, b, c, target, N, P, img_stride, q_lo, q_hi, w0, w1, w2, loss, ws, stream
    "im2im_uq_loss_fwd": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _ptr, _ptr]),
    # kind, a, b, c, target, N, P, img_stride, q_lo, q_hi, w0, w1, w2, grad_out, d_a, d_b, d_c, d_stride, stream
    "im2im_uq_loss_bwd": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "im2im_fastmri_mask_pack": (_i32, [_ptr, _ptr, _i64, _ptr, _i64, _i32, _i32, _i64, _i32, _ptr]),
    "im2im_complex_transpose": (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "im2im_fastmri_abs_normalize": (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _f32, _f32, _ptr]),
    "im2im_center_crop_affine": (_i32, [_ptr, _ptr, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _ptr]),
    "im2im_adam_step": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _f32, _i64, _ptr]),
    "im2im_adam_step_dev": (_i32, [_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _f32, _ptr, _ptr, _ptr]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise Im2ImError(f"{what} failed (rc={rc}): {lib.im2im_last_error().decode()}")


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream (so torch ops and our kernels are stream-ordered)."""
    return torch.cuda.current_stream(device).cuda_stream


def dptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    return t.data_ptr()


def require_gpu(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise Im2ImError(f"{name}: expected a tensor on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise Im2ImError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()

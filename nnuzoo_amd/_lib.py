"""ctypes binding of libnnuzoo_hip.so (include/nnuzoo_hip.h).

The HIP library IS the product: if it is missing or a call fails, this module raises.  There is no CPU or
eager-PyTorch fallback anywhere in the package (the CPU restatements live under /oracle and are test-only).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNZ_HIP_LIBRARY: another build of the same library (experiment builds of tools/probes; same symbols, checked at load)
LIB_PATH = os.environ.get("NNZ_HIP_LIBRARY") or os.path.join(_HERE, "libnnuzoo_hip.so")

NNZ_MAX_GROUPS = 8
NNZ_MAX_TAPS = 32


class ConvTap(C.Structure):
    _fields_ = [("off", C.c_int32 * 3), ("widx", C.c_int32)]


class ConvGroup(C.Structure):
    _fields_ = [("ooff", C.c_int32 * 3), ("tap_begin", C.c_int32), ("ntaps", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("in_dims", C.c_int32 * 3),
        ("out_dims", C.c_int32 * 3),
        ("m_dims", C.c_int32 * 3),
        ("Cin", C.c_int32),
        ("Cout", C.c_int32),
        ("ldi", C.c_int32),
        ("ldo", C.c_int32),
        ("in_stride", C.c_int32 * 3),
        ("out_stride", C.c_int32 * 3),
        ("ext", C.c_int32 * 3),
        ("lo", C.c_int32 * 3),
        ("ntaps_total", C.c_int32),
        ("ngroups", C.c_int32),
        ("accumulate", C.c_int32),
        ("groups", ConvGroup * NNZ_MAX_GROUPS),
        ("taps", ConvTap * NNZ_MAX_TAPS),
    ]


class HipLibraryMissing(ImportError):
    pass


class HipCallError(RuntimeError):
    pass


_vp, _fp, _i, _l, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(ConvDesc)

# name -> argtypes; every symbol declared in include/nnuzoo_hip.h must appear here (tests check both ways)
SIGNATURES = {
    "nnz_version": [],
    "nnz_device_info": [C.c_char_p, _i, _ip, C.POINTER(C.c_long)],
    "nnz_conv_tap_forward": [_vp, _vp, _vp, _fp, _dp, _vp],
    "nnz_conv_tap_forward_stats": [_vp, _vp, _vp, _fp, _dp, _fp, _vp],
    "nnz_conv_tuning": [_i, _i],
    "nnz_conv_tuning_get": [_i],
    "nnz_conv_tap_wgrad": [_vp, _vp, _fp, _dp, _i, _vp],
    "nnz_conv_tap_wgrad_workspace_floats": [_dp],
    "nnz_conv_tap_wgrad_to_grad": [_vp, _vp, _fp, _l, _fp, _l, _l, _l, _ip, _i, _dp, _vp],
    "nnz_pack_conv_weight": [_fp, _vp, _i, _i, _i, _l, _l, _l, _ip, _vp],
    "nnz_pack_job_bytes": [],
    "nnz_pack_job_fill": [_vp, _fp, _vp, _i, _i, _i, _l, _l, _l, _ip],
    "nnz_pack_conv_weights_batched": [_vp, _i, _vp],
    "nnz_pack_dual_job_bytes": [],
    "nnz_pack_dual_job_fill": [_vp, _fp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _ip],
    "nnz_pack_dual_batched": [_vp, _i, _i, _i, _vp],
    "nnz_unpack_conv_wgrad": [_fp, _fp, _i, _i, _i, _l, _l, _l, _ip, _i, _vp],
    "nnz_stem_conv_forward": [_fp, _fp, _fp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_stem_conv_wgrad": [_fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_seg_head_forward": [_vp, _fp, _fp, _vp, _i, _l, _i, _i, _i, _vp],
    "nnz_seg_head_dgrad": [_vp, _fp, _vp, _i, _l, _i, _i, _i, _i, _vp],
    "nnz_seg_head_wgrad": [_vp, _vp, _fp, _fp, _i, _l, _i, _i, _i, _vp],
    "nnz_instnorm_stats": [_vp, _fp, _i, _l, _i, _i, _i, _vp],
    "nnz_instnorm_lrelu_apply": [_vp, _fp, _fp, _fp, _vp, _i, _l, _i, _i, _i, _f, _f, _vp],
    "nnz_instnorm_lrelu_bwd_reduce": [_vp, _vp, _fp, _fp, _fp, _fp, _i, _l, _i, _i, _i, _f, _f, _i, _vp],
    "nnz_instnorm_lrelu_bwd_apply": [_vp, _vp, _fp, _fp, _fp, _fp, _vp, _i, _l, _i, _i, _i, _i, _f, _f, _fp, _fp, _vp],
    "nnz_fxacc_bytes": [],
    "nnz_conv_tap_forward_norm": [_vp, _vp, _vp, _fp, _dp, _vp, _vp, _fp, _fp, _f, _fp, _vp],
    "nnz_conv_tap_forward_ws": [_vp, _vp, _vp, _fp, _dp, _fp, _l, _vp],
    "nnz_conv_tap_forward_norm_ws": [_vp, _vp, _vp, _fp, _dp, _vp, _vp, _fp, _fp, _f, _fp, _fp, _l, _vp],
    "nnz_stem_conv_wgrad_det": [_fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "nnz_seg_head_wgrad_det": [_vp, _vp, _fp, _fp, _i, _l, _i, _i, _i, _vp, _vp, _vp],
    "nnz_dc_ce_loss_forward_det": [_vp, _i, _vp, _fp, _i, _i, _l, _i, _vp, _vp, _vp],
    "nnz_grad_sumsq_nonfinite_det": [_fp, _l, _fp, _vp, _vp, _vp],
    "nnz_instnorm_stats_det": [_vp, _i, _l, _i, _i, _vp, _vp, _fp, _fp, _f, _fp, _fp, _vp],
    "nnz_instnorm_lrelu_apply_tab": [_vp, _fp, _vp, _i, _l, _i, _i, _i, _f, _vp],
    "nnz_conv_tap_dgrad_normred": [_vp, _vp, _vp, _dp, _vp, _i, _fp, _f, _vp, _vp, _fp, _fp, _fp, _vp],
    "nnz_bn_batch_stats_finish": [_fp, _i, _i, _f, _f, _fp, _fp, _fp, _vp],
    "nnz_instnorm_lrelu_bwd_apply_tab": [_vp, _vp, _fp, _fp, _vp, _i, _l, _i, _i, _i, _i, _f, _vp],
    "nnz_instnorm_lrelu_bwd_tab": [_vp, _vp, _fp, _vp, _vp, _fp, _vp, _i, _l, _i, _i, _i, _i, _f, _fp, _fp, _vp],
    "nnz_graph_replace_memsets": [_vp, _ip],
    "nnz_graph_node_census": [_vp, _ip, _i],
    "nnz_convT_supported": [_i] * 6,
    "nnz_dwconv2d_wgrad_workspace_floats": [_i] * 4,
    "nnz_dwconv2d_wgrad": [_fp, _fp, _i, _fp, _fp, _fp] + [_i] * 5 + [_vp],
    "nnz_convT_forward": [_fp, _fp, _fp, _fp] + [_i] * 11 + [_vp],
    "nnz_convT_dgrad": [_fp, _fp, _fp] + [_i] * 11 + [_vp],
    "nnz_adam_chunk_bytes": [],
    "nnz_adamw_fused": [_vp, _i, _fp, _vp, _vp, _fp, _f, _f, C.c_double, C.c_double, _f, _f, _fp, _i, _vp],
    "nnz_conv_tap_forward_innorm": [_vp, _vp, _vp, _fp, _dp, _fp, _i, _f, _vp, _vp, _fp, _fp, _f, _fp, _fp, _l, _vp],
    "nnz_conv_tap_wgrad_to_grad_innorm": [_vp, _vp, _fp, _l, _fp, _l, _l, _l, _ip, _i, _dp, _fp, _i, _f, _fp, _i, _f, _vp],
    "nnz_seg_head_forward_innorm": [_vp, _fp, _f, _fp, _fp, _vp, _i, _l, _i, _i, _i, _vp],
    "nnz_seg_head_wgrad_innorm": [_vp, _fp, _f, _vp, _fp, _fp, _i, _l, _i, _i, _i, _vp, _vp, _vp],
    "nnz_convT_forward_innorm": [_fp, _fp, _f, _fp, _fp, _fp] + [_i] * 11 + [_vp],
    "nnz_crop_pad_f32": [_vp, _ip, _ip, _ip, _fp, _i, _i, _i, _i, _i, _f, _vp],
    "nnz_crop_pad_i16": [_vp, _ip, _ip, _ip, _fp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_downsample_nearest_i16": [_fp, _fp, _l, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_aug_affine_f32": [_fp, _fp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "nnz_aug_affine_i16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_aug_stats_workspace_floats": [_i],
    "nnz_aug_stats_f32": [_fp, _l, _i, _fp, _fp, _vp],
    "nnz_aug_intensity_f32": [_fp, _l, _i, _i, _fp, _fp, _fp, _i, _vp],
    "nnz_aug_relabel_i16": [_vp, _l, _i, _i, _vp],
    "nnz_aug_seg_to_regions_i16": [_vp, _vp, _i, _i, _i, _l, _vp, _vp, _i, _vp],
    "nnz_aug_seg_onehot_to_data_f32": [_vp, _vp, _i, _i, _i, _i, _i, _l, _vp, _i, _vp],
    "nnz_aug_mask_outside_f32": [_vp, _vp, _i, _i, _i, _i, _l, _l, _f, _vp],
    "nnz_aug_blur_axis_f32": [_fp, _fp, _i, _i, _i, _i, _i, _fp, _vp],
    "nnz_aug_lowres_f32": [_fp, _fp, _i, _i, _i, _i, _i, _fp, _vp],
    "nnz_ss2d_xproj_forward": [_fp, _fp, _fp, _i, _i, _i, _l, _i, _vp],
    "nnz_ss2d_xproj_backward_x": [_fp, _fp, _fp, _fp, _i, _i, _i, _l, _i, _vp],
    "nnz_ss2d_xproj_backward_w": [_fp, _fp, _fp, _i, _i, _i, _l, _i, _vp],
    "nnz_ss2d_xproj_backward_w_ws": [_fp, _fp, _fp, _fp, _l, _i, _i, _i, _l, _i, _vp],
    "nnz_ss2d_xproj_backward_w_workspace_floats": [_i, _i, _i, _l],
    "nnz_token_linear_forward": [_vp, _fp, _fp, _vp, _l, _i, _i, _i, _vp],
    "nnz_token_linear_supported": [_i, _i],
    "nnz_dense32_forward": [_fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _vp],
    "nnz_dense32_dgrad": [_fp, _fp, _fp, _fp, _l, _i, _i, _vp],
    "nnz_dense32_wgrad_workspace_floats": [_l, _i, _i],
    "nnz_dense32_wgrad": [_fp, _fp, _fp, _fp, _fp, _l, _i, _i, _vp],
    "nnz_dense32_group_record_bytes": [_i],
    "nnz_dense32_group_plan": [_l, _i, _i, _vp, _vp, _vp],
    "nnz_dense32_group_fill": [_vp, _vp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _i],
    "nnz_dense32_group_class": [_l, _i, _i],
    "nnz_dense32_group_launch": [_vp, _vp, _i, _vp, _vp, _i, _i, _vp],
    "nnz_group_fold_launch": [_vp, _vp, _i, _vp],
    "nnz_token_linear_wgrad_group_record_bytes": [],
    "nnz_token_linear_wgrad_group_plan": [_l, _i, _i, _vp, _vp, _vp],
    "nnz_token_linear_wgrad_group_fill": [_vp, _vp, _vp, _vp, _l, _i, _i, _i],
    "nnz_token_linear_wgrad_group_launch": [_vp, _vp, _i, _i, _vp],
    "nnz_ss2d_xproj_backward_w_group_record_bytes": [],
    "nnz_ss2d_xproj_backward_w_group_plan": [_i, _i, _i, _l, _vp, _vp, _vp],
    "nnz_ss2d_xproj_backward_w_group_fill": [_vp, _fp, _fp, _vp, _i, _i, _i, _l, _i, _i],
    "nnz_ss2d_xproj_backward_w_group_launch": [_vp, _vp, _i, _i, _vp],
    "nnz_dense32_splitk_workspace_floats": [_l, _i, _i],
    "nnz_dense32_forward_fused": [_fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _fp, _fp, _f, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp,
                                  _f, _i, _i, _fp, _vp],
    "nnz_dense32_dgrad_fused": [_fp, _fp, _fp, _fp, _l, _i, _i, _fp, _f, _i, _i, _fp, _vp],
    "nnz_dense32_group_fill_scaled": [_vp, _vp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _i, _fp, _f, _i, _i],
    "nnz_dense32_group_fill_fold": [_vp, _fp, _fp, _l, _i, _i],
    "nnz_dense32_forward_h16": [_vp, _fp, _fp, _vp, _l, _i, _i, _fp, _vp],
    "nnz_dense32_dgrad_h16": [_vp, _fp, _vp, _l, _i, _i, _fp, _vp],
    "nnz_dense32_group_fill_h16": [_vp, _vp, _vp, _vp, _fp, _fp, _fp, _l, _i, _i, _i, _i],
    "nnz_window_attention_forward_pad": [_fp, _fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "nnz_window_attention_backward_pad": [_fp, _fp, _vp, _fp, _fp, _fp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "nnz_window_attention_backward_parts": [_i, _i, _i, _i],
    "nnz_window_attention_backward_partial": [_fp, _fp, _vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "nnz_layer_norm_backward_parts": [_l, _i],
    "nnz_layer_norm_backward_partial": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _i, _i, _vp],
    "nnz_token_linear_wgrad": [_vp, _vp, _fp, _fp, _l, _i, _i, _vp],
    "nnz_token_linear_wgrad_ws": [_vp, _vp, _fp, _fp, _fp, _l, _l, _i, _i, _vp],
    "nnz_token_linear_wgrad_workspace_floats": [_l, _i, _i],
    "nnz_sgd_chunk_bytes": [],
    "nnz_sgd_chunk_fill": [_vp, _vp, _vp, _l, _i],
    "nnz_grad_sumsq_nonfinite": [_fp, _l, _fp, _vp],
    "nnz_sgd_nesterov_fused": [_vp, _i, _fp, _fp, _fp, _f, _f, _f, _f, _i, _vp],
    "nnz_dc_bce_loss_forward": [_vp, _i, _vp, _fp, _i, _i, _i, _l, _vp],
    "nnz_dc_bce_loss_backward": [_vp, _i, _vp, _fp, _vp, _i, _i, _i, _l, _vp],
    "nnz_region_tp_fp_fn": [_vp, _i, _vp, _vp, _i, _i, _i, _l, _vp],
    "nnz_argmax_tp_fp_fn": [_vp, _i, _vp, _vp, _i, _i, _l, _i, _vp],
    "nnz_ss2d_prepare": [_vp, _i, _fp, _i, _i, _i, _i, _vp],
    "nnz_ss2d_merge": [_fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_ss2d_split": [_fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_ss2d_merge_dx": [_fp, _fp, _vp, _i, _i, _i, _i, _i, _vp],
    "nnz_ss2d_dwconv_silu_forward": [_vp, _i, _l, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_ss2d_dwconv_silu_backward": [_vp, _i, _l, _fp, _fp, _fp, _vp, _fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_ss2d_dwconv_silu_backward_ws": [_vp, _i, _l, _fp, _fp, _fp, _vp, _fp, _fp, _fp, _l, _i, _i, _i, _i, _vp],
    "nnz_ss2d_dwconv_silu_backward_workspace_floats": [_i, _i, _i, _i],
    "nnz_ss2d_scan_state_floats": [_i, _i, _i],
    "nnz_ss2d_scan_grad_state_floats": [_i, _i, _i],
    "nnz_ss2d_scan_workspace_floats": [_i, _i, _i],
    "nnz_scan_tuning": [_i, _i],
    "nnz_norm_tuning": [_i, _i],
    "nnz_scan_tuning_get": [_i],
    "nnz_ss2d_scan_forward": [_fp] * 9 + [_i, _i, _i, _i, _i, _i, _vp],
    "nnz_ss2d_scan_backward": [_fp] * 16 + [_i, _i, _i, _i, _i, _i, _vp],
    "nnz_residual_droppath_forward": [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _l, _vp],
    "nnz_residual_droppath_backward": [_vp, _i, _vp, _i, _f, _vp, _i, _i, _l, _vp],
    "nnz_residual_droppath_rand_forward": [_vp, _i, _vp, _i, _fp, _f, _f, _vp, _i, _i, _l, _vp],
    "nnz_residual_droppath_rand_backward": [_vp, _i, _fp, _f, _f, _vp, _i, _i, _l, _vp],
    "nnz_dw3x3_nhwc_f32": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp],
    "nnz_dw3x3_nhwc_wgrad_workspace_floats": [_i, _i, _i, _i],
    "nnz_dw3x3_nhwc_wgrad_f32": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_bn_relu_nhwc_forward_f32": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _f, _f, _vp],
    "nnz_bn_relu_nhwc_backward_f32": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _vp],
    "nnz_pw_wgrad_small_workspace_floats": [_l, _i, _i],
    "nnz_pw_wgrad_small_f32": [_fp, _fp, _fp, _fp, _l, _i, _i, _vp],
    "nnz_pw_wgrad_small_workspace_floats_b": [_l, _i, _i, _i],
    "nnz_pw_wgrad_small": [_vp, _vp, _i, _fp, _fp, _fp, _l, _i, _i, _vp],
    "nnz_head1x1_forward_f32": [_fp, _fp, _fp, _fp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_head1x1_dgrad_f32": [_fp, _fp, _fp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_head1x1_wgrad_workspace_floats": [_i, _i, _i, _l],
    "nnz_head1x1_wgrad_f32": [_fp, _fp, _fp, _fp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_bilinear_up_forward": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_bilinear_up_backward": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_head1x1_forward_f16": [_vp, _fp, _fp, _vp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_head1x1_dgrad_f16": [_vp, _fp, _vp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_head1x1_wgrad_f16": [_vp, _vp, _fp, _fp, _i, _i, _i, _l, _l, _l, _l, _vp],
    "nnz_pad_top_left": [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_crop_top_left": [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp],
    "nnz_layer_norm_gate_forward": [_vp, _i, _fp, _fp, _vp, _i, _l, _vp, _i, _fp, _fp, _fp, _l, _i, _f, _vp],
    "nnz_layer_norm_gate_backward": [_vp, _i, _fp, _fp, _vp, _i, _l, _fp, _fp, _vp, _i, _vp, _vp, _fp, _fp, _i, _l, _i,
                                     _vp],
    "nnz_layer_norm_forward": [_vp, _i, _fp, _fp, _vp, _i, _fp, _fp, _fp, _l, _i, _f, _vp],
    "nnz_layer_norm_backward": [_vp, _i, _fp, _fp, _fp, _vp, _i, _vp, _fp, _fp, _i, _l, _i, _vp],
    "nnz_layer_norm_backward_det": [_vp, _i, _fp, _fp, _fp, _vp, _i, _vp, _fp, _fp, _vp, _vp, _l, _i, _vp],
    "nnz_layer_norm_backward_det_res": [_vp, _i, _fp, _fp, _fp, _vp, _i, _fp, _vp, _fp, _fp, _vp, _vp, _l, _i, _vp],
    "nnz_layer_norm_gate_backward_det": [_vp, _i, _fp, _fp, _vp, _i, _l, _fp, _fp, _vp, _i, _vp, _vp, _fp, _fp, _vp, _vp, _l,
                                         _i, _vp],
    "nnz_dc_ce_loss_forward": [_vp, _i, _vp, _fp, _i, _i, _l, _i, _vp],
    "nnz_dc_ce_loss_backward": [_vp, _i, _vp, _fp, _vp, _i, _i, _l, _i, _vp],
    "nnz_dc_ce_loss_finalize": [_fp, _fp, _fp, _i, _i, _l, _i, _i, _f, _f, _f, _f, _i, _vp],
    "nnz_dc_ce_loss_backward_scaled": [_vp, _i, _vp, _fp, _fp, _vp, _i, _i, _l, _i, _vp],
    "nnz_sliding_window_accumulate": [_vp, _i, _ip, _vp, _vp, _vp, _i, _ip, _ip, _ip, _vp],
    "nnz_sliding_window_finalize": [_vp, _vp, _i, _l, _vp, _vp],
    "nnz_causal_conv1d_silu_forward": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_causal_conv1d_silu_backward": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp],
    "nnz_silu_gate_forward": [_fp, _fp, _fp, _l, _vp],
    "nnz_silu_gate_backward": [_fp, _fp, _fp, _fp, _fp, _l, _vp],
    "nnz_global_attention_forward": [_fp, _fp, _fp, _i, _i, _i, _i, _f, _vp],
    "nnz_global_attention_backward": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _vp],
    "nnz_window_attention_forward": [_fp, _fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "nnz_window_attention_backward": [_fp, _fp, _vp, _fp, _fp, _fp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "nnz_selective_scan_workspace_floats": [_i, _i, _i],
    "nnz_selective_scan_state_floats": [_i, _i, _i],
    "nnz_selective_scan_grad_state_floats": [_i, _i, _i],
    "nnz_selective_scan_forward": [_fp] * 10 + [_i] * 6 + [_vp],
    "nnz_selective_scan_backward": [_fp] * 18 + [_i] * 6 + [_vp],
}

_LONG_RESULT = {"nnz_aug_stats_workspace_floats", "nnz_ss2d_scan_state_floats", "nnz_ss2d_scan_grad_state_floats", "nnz_ss2d_scan_workspace_floats",
                "nnz_selective_scan_workspace_floats", "nnz_selective_scan_state_floats",
                "nnz_selective_scan_grad_state_floats", "nnz_dwconv2d_wgrad_workspace_floats", "nnz_conv_tap_wgrad_workspace_floats",
                "nnz_dense32_wgrad_workspace_floats", "nnz_token_linear_wgrad_workspace_floats",
                "nnz_dense32_splitk_workspace_floats", "nnz_layer_norm_backward_parts",
                "nnz_window_attention_backward_parts",
                "nnz_ss2d_xproj_backward_w_workspace_floats", "nnz_ss2d_dwconv_silu_backward_workspace_floats",
                "nnz_dw3x3_nhwc_wgrad_workspace_floats", "nnz_pw_wgrad_small_workspace_floats",
                "nnz_head1x1_wgrad_workspace_floats", "nnz_pw_wgrad_small_workspace_floats_b"}
_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises HipLibraryMissing with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m nnuzoo_amd.build` (hipcc --offload-arch=gfx950). "
            "nnuzoo_amd has no CPU fallback.")
    # torch first: PyTorch-ROCm ships its own libamdhip64; our library's NEEDED entry must resolve to that already-loaded
    # runtime.  Loaded the other way round the process holds two HIP runtimes and every launch from here fails with
    # hipErrorNoDevice (seen when a test touched the library before anything imported torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = C.c_long if name in _LONG_RESULT else C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "invalid argument / unsupported shape" if rc == -22 else f"hipError_t {rc}"
        raise HipCallError(f"{what} failed: {kind}")


def call(name: str, *args) -> None:
    lib = load()
    check(getattr(lib, name)(*args), name)


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """data_ptr of a CUDA tensor (None -> NULL)."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise HipCallError("nnuzoo_amd kernels need device (HIP) tensors; there is no CPU path")
    return t.data_ptr()

"""ctypes binding of libpaif_hip.so (C ABI: include/paif_hip.h).

The library is REQUIRED: there is no CPU or eager-PyTorch fallback anywhere in paif_amd.  If the
shared object is missing or a symbol is absent, importing an op raises immediately.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_size_t, c_ulonglong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAIF_LIB") or os.path.join(HERE, "lib", "libpaif_hip.so")  # PAIF_LIB: A/B builds of the same ABI

F = c_void_p  # device pointer


class ConvDesc(Structure):
    """paif_conv_desc (include/paif_hip.h)."""

    _fields_ = [
        ("src", F * 3), ("nsrc", c_int), ("cin", c_int), ("wpk", F), ("kh", c_int), ("dil", c_int),
        ("in_act", c_int), ("in_prelu", F), ("scale", F), ("shift", F), ("act", c_int), ("prelu", F),
        ("alpha", c_float), ("res", F * 3), ("out", F), ("cout", c_int), ("pool_partial", F),
        ("precision", c_int), ("aux_out", F), ("in_aux", F), ("in_scale", F), ("in_alpha", c_float), ("epi_aux", F),
        ("epi_dact", c_int), ("reverse_tiles", c_int), ("storage", c_int), ("cpool", F),
    ]


# name -> (restype, argtypes); every symbol declared in include/paif_hip.h
SIGNATURES = {
    "paif_version": (c_int, []),
    "paif_last_error": (c_char_p, []),
    "paif_device_cus": (c_int, []),
    "paif_rgb2ycrcb_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_ycrcb2rgb_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_minmax_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_recompose_clamp_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, F]),
    "paif_fused_uint8_fwd": (c_int, [F, F, c_int, F, c_int, c_int, c_int, F]),
    "paif_minmax_normalize_fwd": (c_int, [F, F, c_int, F, F, c_int, c_int, c_int, F]),
    "paif_stem_fwd": (c_int, [F, c_size_t, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_stem_fwd_twin": (c_int, [F, c_size_t, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_stem_fwd_twin_f16": (c_int, [F, c_size_t, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_channel_residue_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_ab_fwd": (c_int, [F, F, F, c_float, c_float, c_int, c_int, c_int, F]),
    "paif_guided_filter_lf_fwd": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_taped_fits": (c_int, [c_int, c_int, c_int]),
    "paif_guided_filter_taped_fwd": (c_int, [F, F, F, F, c_float, c_float, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_fused_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "paif_guided_filter_fused_fwd": (c_int, [F, F, F, c_float, c_float, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_fused_fwd_bf16": (c_int, [F, F, F, c_float, c_float, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_fused_fwd_hf16": (c_int, [F, F, F, c_float, c_float, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_fused_fwd_hf16_y16": (c_int, [F, F, F, c_float, c_float, F, c_int, c_int, c_int, F]),
    "paif_conv2d_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_conv2d_fwd": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int, F]),
    "paif_conv2d_is_persistent": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int]),
    "paif_conv2d_can_cpool": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int]),
    "paif_rdb_fused_wpk_floats": (c_size_t, []),
    "paif_rdb_fused_pack": (c_int, [F, F, F, F, c_int, F]),
    "paif_rdb_fused_fwd": (c_int, [F, F, F, c_float, F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_channel_pool1_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_channel_pool1_fwd_bf16": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_channel_pool1_fwd_f16": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_conv2d_kernel_name": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int, c_char_p, c_int]),
    "paif_conv_wpk_floats": (c_size_t, [c_int, c_int, c_int]),
    "paif_pack_conv_weight": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_pack_decomp1x1_weight": (c_int, [F, F, F]),
    "paif_pack_conv_weight_bf16x3": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_pack_conv_weight_f16x2": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_pack_conv_weight_bf16x6": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_compose_dw_pw_weight": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_compose_pw_conv_weight": (c_int, [F, F, F, c_int, c_int, c_int, c_int, F]),
    "paif_stem_out_pack_floats": (c_int, []),
    "paif_stem_out_pack": (c_int, [F, F, F, F]),
    "paif_stem_out_fwd_bf16": (c_int, [F, F, F, F, c_int, c_int, c_int, F]),
    "paif_stem_out_fwd_f32": (c_int, [F, F, F, F, c_int, c_int, c_int, F]),
    "paif_stem_out_fwd_guard": (c_int, [F, c_int, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_pack_decomp1x1_weight_bf16x3": (c_int, [F, F, F]),
    "paif_pack_decomp1x1_hf_weight_f16x2": (c_int, [F, F, F]),
    "paif_pack_decomp1x1_weight_bf16x6": (c_int, [F, F, F]),
    "paif_bn_fold": (c_int, [F, F, F, F, c_float, F, F, c_int, F]),
    "paif_dwconv_fwd": (c_int, [F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_channel_pool2_fwd": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_spa_blend_fwd": (c_int, [F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_eca_finish_fwd": (c_int, [F, F, F, F, c_int, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_tail_fwd": (c_int, [F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_add_fwd": (c_int, [F, F, F, c_size_t, F]),
    "paif_gemm_fwd": (c_int, [F, c_int, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_gemm_splitk_plan": (c_int, [c_int, c_int, c_int]),
    "paif_gemm_conv_fwd": (c_int, [F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, F, F]),
    "paif_gemm_col2im_fwd": (c_int, [F, c_int, F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_timing_event_create": (c_int, [POINTER(c_void_p)]),
    "paif_timing_event_record": (c_int, [c_void_p, F]),
    "paif_timing_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "paif_timing_event_destroy": (c_int, [c_void_p]),
    "paif_gemm2_plan": (c_int, [c_int, c_int, c_int, c_int]),
    "paif_gemm2_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "paif_gemm2_pack_weight": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_gemm2_fwd": (c_int, [F, c_int, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_conv2d_wgrad_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "paif_conv2d_wgrad": (c_int, [POINTER(c_void_p), c_int, F, F, F, F, c_int, c_float, c_int, c_int, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_gemm_wgrad_splits": (c_int, [c_int, c_int, c_int]),
    "paif_gemm_wgrad": (c_int, [F, c_int, F, c_int, F, F, c_int, c_int, c_int, c_int, F, c_int, F]),
    "paif_gemm_wgrad_p": (c_int, [F, c_int, F, c_int, F, F, c_int, c_int, c_int, c_int, F, c_int, c_int, c_float, F]),
    "paif_layernorm_wgrad_blocks": (c_int, [c_int]),
    "paif_layernorm_wgrad": (c_int, [F, F, F, F, F, c_int, c_int, c_float, c_int, F]),
    "paif_channel_affine_nchw_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, F]),
    "paif_image_loss_blocks": (c_int, []),
    "paif_image_loss_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, F, F, F]),
    "paif_image_loss_bwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_float, F, F]),
    "paif_ssim_l1_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_ssim_l1_fwd": (c_int, [F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_ssim_l1_bwd_input": (c_int, [F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_gemm_splitk_fwd": (c_int, [F, c_int, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, c_int, F, F]),
    "paif_gemm_splitk_fwd_p": (c_int, [F, c_int, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, c_int, F, c_int, F]),
    "paif_layernorm_fwd": (c_int, [F, F, F, F, c_int, c_int, c_float, F]),
    "paif_im2col_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_pack_conv_gemm_weight": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_dwconv3_bias_gelu_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, F]),
    "paif_sr_attention_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_sr_attention_bf16x3_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_sr_attention_split_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_resize_bilinear_into_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_nhwc_to_nchw_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_nchw_to_nhwc_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_nchw_to_nhwc_pad_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_gemm_masked_fwd": (c_int, [F, c_int, F, F, F, F, F, c_int, F, c_int, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_transpose_pad_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_layernorm_bwd_input": (c_int, [F, F, F, F, F, c_int, c_int, c_float, F]),
    "paif_dwconv3_bias_gelu_bwd_input": (c_int, [F, F, F, F, F, F, c_int, c_int, c_int, c_int, F]),
    "paif_col2im_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_resize_bilinear_adjoint_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_sr_attention_bwd_chunks": (c_int, [c_int, c_int, c_int]),
    "paif_sr_attention_bwd_input": (c_int, [F, F, F, F, F, F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_sr_attention_bwd_input_p": (c_int, [F, F, F, F, F, F, F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_upsample_ce_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_upsample_ce_fwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_dwconv_fwd_bf16": (c_int, [F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_channel_pool2_fwd_bf16": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_spa_blend_fwd_bf16": (c_int, [F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_eca_finish_fwd_bf16": (c_int, [F, F, F, F, c_int, F, F, F, c_int, c_int, c_int, F]),
    "paif_tail_fwd_bf16": (c_int, [F, F, F, F, c_int, c_int, c_int, F]),
    "paif_add_fwd_bf16": (c_int, [F, F, F, c_size_t, F]),
    "paif_dwconv_fwd_f16": (c_int, [F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_channel_pool2_fwd_f16": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_spa_blend_fwd_f16": (c_int, [F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_eca_finish_fwd_f16": (c_int, [F, F, F, F, c_int, F, F, F, c_int, c_int, c_int, F]),
    "paif_tail_fwd_f16": (c_int, [F, F, F, F, c_int, c_int, c_int, F]),
    "paif_add_fwd_f16": (c_int, [F, F, F, c_size_t, F]),
    "paif_cast_storage_fwd": (c_int, [F, F, c_size_t, c_int, F]),
    "paif_attack_loss_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_attack_loss_fwd": (c_int, [F, F, F, F, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_attack_loss_bwd": (c_int, [F, F, F, F, c_int, c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_conv_weight_dgrad": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_fold_decomp1x1_weight": (c_int, [F, F, F]),
    "paif_tail_bwd_input": (c_int, [F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_stem_bwd_input": (c_int, [F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_dwconv_bwd_input": (c_int, [F, F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_eca_bwd_blocks": (c_int, [c_int, c_int]),
    "paif_eca_bwd_input": (c_int, [F, F, F, F, F, c_int, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_spa_blend_bwd_input": (c_int, [F, F, F, F, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_bwd_input": (c_int, [F, F, F, F, c_float, c_float, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_guided_filter_bwd_input_mc": (c_int, [F, F, F, F, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_upsample_argmax_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_confusion_matrix_accum": (c_int, [F, F, F, c_size_t, c_int, F]),
    "paif_spa1_fwd": (c_int, [F, F, F, c_int, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_spa1_bwd_input": (c_int, [F, F, F, F, F, c_int, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_glue_bwd_blocks": (c_int, [c_int, c_int, c_int]),
    "paif_glue_bwd_input": (c_int, [F, F, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_plane_minmax_blocks": (c_int, [c_size_t]),
    "paif_plane_clamp_minmax_fwd": (c_int, [F, F, F, F, c_size_t, F]),
    "paif_plane_clamp_minmax_bwd_input": (c_int, [F, F, F, F, F, c_size_t, F]),
    "paif_channel_sum_chunks_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_rgb2ycrcb_bwd_input": (c_int, [F, F, F, c_int, c_int, c_int, F]),
    "paif_pgd_step": (c_int, [F, F, F, c_float, c_float, c_size_t, F]),
    "paif_axpy": (c_int, [F, F, c_float, c_size_t, F]),
    "paif_upsample_ce_bwd": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_head_sum_fwd": (c_int, [F, F, F, F, POINTER(c_int), F, F, F, c_int, c_int, F]),
    "paif_relu_mask_scale_fwd": (c_int, [F, F, F, F, c_size_t, c_int, F]),
    "paif_u8_to_planes_fwd": (c_int, [F, F, c_int, c_int, c_int, F]),
    "paif_u8_to_i64_fwd": (c_int, [F, F, c_size_t, F]),
    # ---- training step (csrc/train_kernels.hip) ----
    "paif_nhwc_slice_to_nchw_fwd": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_pad_channels_fwd": (c_int, [F, F, c_size_t, c_int, c_int, F]),
    "paif_decomp_cat_fwd": (c_int, [F, F, F, F, c_size_t, F]),
    "paif_row_reduce_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "paif_bn_stats_fwd": (c_int, [F, c_int, c_int, F, F, c_float, c_float, F, F, F, F, F, F, F, F]),
    "paif_affine_act_res_fwd": (c_int, [F, F, F, c_int, F, F, F, F, F, c_size_t, c_int, F]),
    "paif_bn_act_bwd": (c_int, [F, F, F, F, F, F, c_int, F, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_bn_eval_stats": (c_int, [F, F, F, F, c_float, c_int, F, F, F, F, F]),
    "paif_prelu_bwd": (c_int, [F, F, F, F, c_float, F, F, F, c_size_t, F]),
    "paif_tail_dz": (c_int, [F, F, F, F, F, F, F, c_size_t, F]),
    "paif_colsum": (c_int, [F, c_int, F, F, c_int, c_int, F]),
    "paif_dwconv_wgrad": (c_int, [F, F, F, F, F, c_int, c_int, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_stem_wgrad": (c_int, [F, c_size_t, F, F, F, F, F, F, c_int, c_int, c_int, F]),
    "paif_corr1_wgrad": (c_int, [F, F, F, F, c_int, c_int, c_int, c_int, c_int, F]),
    "paif_eca_wgrad": (c_int, [F, F, c_int, F, c_int, F, F, c_int, c_int, c_int, F]),
    "paif_unfold_decomp1x1_wgrad": (c_int, [F, F, F]),
    "paif_unpack_conv_gemm_wgrad": (c_int, [F, F, c_int, c_int, c_int, c_int, F]),
    "paif_keep_mask": (c_int, [F, c_int, c_ulonglong, c_ulonglong, c_float, F]),
    "paif_rowscale_add_fwd": (c_int, [F, F, F, F, c_int, c_size_t, c_int, c_int, F]),
    "paif_adamw_step": (c_int, [F, F, F, F, F, c_size_t, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float, c_float, c_float,
                                c_float, F]),
}

_lib = None


class PaifLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Fails loudly -- never falls back to another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64.  Loaded after this library (which would pull /opt/rocm's copy in), a process
    # ends up with two HIP runtimes and kernel launches fail with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PaifLibraryError(
            "libpaif_hip.so not found at %s -- build it with `python -m paif_amd.build` "
            "(hipcc --offload-arch=gfx950).  paif_amd has no CPU/eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PaifLibraryError("libpaif_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().paif_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))

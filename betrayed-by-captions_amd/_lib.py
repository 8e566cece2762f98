"""ctypes binding of libcgg_hip.so (the C ABI declared in include/cgg_hip.h).

There is NO fallback: if the shared library is missing, or a wrapped op is handed tensors that do not
live on a ROCm device, the call raises. The CPU restatement under /oracle is test infrastructure and
is never imported from here.
"""
import ctypes
import os

import torch  # noqa: F401  (imported first so libamdhip64.so.7 resolves to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libcgg_hip.so')

CGG_F32 = 0
CGG_BF16 = 1

_c_int = ctypes.c_int
_c_vp = ctypes.c_void_p
_c_f = ctypes.c_float
_c_i64 = ctypes.c_int64

# name -> (restype, argtypes); must list every symbol of include/cgg_hip.h
PROTOTYPES = {
    'cgg_version': (_c_int, []),
    'cgg_init': (_c_int, [_c_int]),
    'cgg_msda_forward_fused_vld': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 7 + [_c_vp]),
    'cgg_msda_backward_hostlevels_ws': (_c_int, [_c_vp, _c_int] + [_c_vp] * 8 + [_c_int] * 8 + [_c_vp, _c_i64, _c_vp, _c_vp]),
    'cgg_msda_backward_workspace_bytes': (_c_i64, [_c_vp, _c_vp] + [_c_int] * 7),
    'cgg_msda_read_levels': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    'cgg_last_error_string': (ctypes.c_char_p, []),
    'cgg_msda_forward': (_c_int, [_c_vp] * 6 + [_c_int] * 8 + [_c_vp]),
    'cgg_msda_forward_fused': (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 2 + [_c_int] * 8 + [_c_vp]),
    'cgg_msda_forward_hostlevels': (_c_int, [_c_vp] * 6 + [_c_int] + [_c_vp] + [_c_int] * 9 + [_c_vp]),
    'cgg_msda_backward': (_c_int, [_c_vp] * 9 + [_c_int] * 7 + [_c_vp]),
    'cgg_msda_backward_hostlevels': (_c_int, [_c_vp] * 9 + [_c_int] * 8 + [_c_vp]),
    'cgg_msda_backward_hostlevels_2s': (_c_int, [_c_vp] * 9 + [_c_int] * 8 + [_c_vp, _c_vp]),
    'cgg_msda_backward_overwrites': (_c_int, [_c_vp] * 2 + [_c_int] * 7),
    'cgg_pack_mask_feature': (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    'cgg_mask_logits': (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    'cgg_mask_logits_bits_astat': (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_mask_logits_f32': (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
    'cgg_mask_logits_backward_workspace_bytes': (_c_i64, [_c_int] * 4),
    'cgg_mask_logits_backward': (_c_int, [_c_vp] * 6 + [_c_int] * 5 + [_c_vp]),
    'cgg_attn_mask_fix_full_rows': (_c_int, [_c_vp, _c_int, _c_int, _c_vp]),
    'cgg_attn_mask_from_logits': (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_masked_xattn_workspace_bytes': (_c_i64, [_c_int] * 5),
    'cgg_masked_xattn_forward': (_c_int, [_c_vp] * 5 + [_c_int] * 5 + [_c_f, _c_int, _c_vp]),
    'cgg_masked_xattn_forward_strided': (_c_int, [_c_vp, _c_vp, _c_int, _c_i64, _c_vp, _c_vp, _c_vp] + [_c_int] * 5 + [_c_f, _c_vp]),
    'cgg_masked_xattn_forward_x3': (_c_int, [_c_vp, _c_vp, _c_int, _c_i64, _c_vp, _c_vp, _c_vp] + [_c_int] * 5 + [_c_f, _c_vp]),
    'cgg_masked_xattn_forward_lse': (_c_int, [_c_vp] * 6 + [_c_int] * 5 + [_c_f, _c_int, _c_vp]),
    'cgg_masked_xattn_backward_workspace_bytes': (_c_i64, [_c_int] * 5),
    'cgg_masked_xattn_backward': (_c_int, [_c_vp] * 9 + [_c_int] * 5 + [_c_f, _c_int, _c_vp]),
    'cgg_masked_xattn_backward_x3': (_c_int, [_c_vp] * 10 + [_c_int] * 5 + [_c_f, _c_vp]),
    'cgg_grounding_pair_costs': (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_f, _c_vp]),
    'cgg_grounding_pair_costs_backward': (_c_int, [_c_vp] * 5 + [_c_int] * 5 + [_c_f, _c_vp]),
    'cgg_ce_rows_forward': (_c_int, [_c_vp] * 4 + [_c_int, _c_int, _c_i64, _c_i64, _c_int, _c_vp]),
    'cgg_ce_rows_backward': (_c_int, [_c_vp] * 4 + [_c_int, _c_int, _c_i64, _c_i64, _c_int, _c_vp]),
    'cgg_match_cost_rows': (_c_int, [_c_vp] * 4 + [_c_int, _c_int, _c_int, _c_vp]),
    'cgg_rle_encode_bitmasks': (_c_i64, [_c_vp, _c_int, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_i64, _c_vp]),
    'cgg_masked_xattn_forward_bf16': (_c_int, [_c_vp] * 6 + [_c_int] * 5 + [_c_f, _c_int, _c_i64, _c_int, _c_vp]),
    'cgg_linear_rows': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp] + [_c_int] * 6 + [_c_vp]),
    'cgg_add_layernorm': (_c_int, [_c_vp] * 5 + [_c_int, _c_int, _c_f, _c_vp]),
    'cgg_linear_rows_packed_bytes': (_c_i64, [_c_int, _c_int]),
    'cgg_linear_rows_pack': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_linear_rows_bf16': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_f,
                                      _c_vp, _c_int, _c_vp, _c_int] + [_c_int] * 5 +
                             [_c_vp, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_decoder_tail_bf16': (_c_int, [_c_vp, _c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_int, _c_vp, _c_vp, _c_f] +
                              [_c_vp] * 12 + [_c_int, _c_int, _c_vp]),
    'cgg_self_attn_rows_bf16': (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp] + [_c_int] * 4 + [_c_f, _c_vp]),
    'cgg_decoder_mid_bf16': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_int] +
                             [_c_vp] * 5 + [_c_int, _c_int, _c_vp]),
    'cgg_decoder_ffn_bf16': (_c_int, [_c_vp, _c_int] + [_c_vp] * 5 + [_c_int] * 3 + [_c_vp]),
    'cgg_x3_packed_bytes': (_c_i64, [_c_int, _c_int]),
    'cgg_x3_pack': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_linear_rows_x3': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_f,
                                    _c_vp, _c_int, _c_vp, _c_int] + [_c_int] * 5 +
                           [_c_vp, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_decoder_tail_x3': (_c_int, [_c_vp, _c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_int, _c_vp, _c_vp, _c_f] +
                            [_c_vp] * 12 + [_c_int, _c_int, _c_vp]),
    'cgg_decoder_mid_x3': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_int] +
                           [_c_vp] * 5 + [_c_int, _c_int, _c_vp]),
    'cgg_decoder_ffn_x3': (_c_int, [_c_vp, _c_int] + [_c_vp] * 5 + [_c_int] * 3 + [_c_vp]),
    'cgg_encoder_layer_tail_x3': (_c_int, [_c_vp] * 6 + [_c_f] + [_c_vp] * 6 + [_c_f, _c_vp, _c_int, _c_vp, _c_vp] +
                                  [_c_int] * 3 + [_c_vp]),
    'cgg_gemm_x3': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_int] * 4 + [_c_vp]),
    'cgg_stem_conv7x7_x3_nchw': (_c_int, [_c_vp] * 4 + [_c_int] * 3 + [_c_vp]),
    'cgg_bias_relu_maxpool_nhwc_f32': (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_gemm_x3_ex': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_int, _c_vp, _c_int, _c_int] +
                       [_c_int] * 4 + [_c_vp]),
    'cgg_conv_x3_nhwc': (_c_int, [_c_vp] * 5 + [_c_int] * 10 + [_c_vp]),
    'cgg_gemm_x3s': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_int] + [_c_int] * 4 + [_c_vp]),
    'cgg_gemm_x3s_batched': (_c_int, [_c_vp, _c_int, _c_int, _c_i64, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_int] + [_c_int] * 4 + [_c_vp]),
    'cgg_conv_x3s_nhwc': (_c_int, [_c_vp] * 4 + [_c_int, _c_vp, _c_int] + [_c_int] * 10 + [_c_vp]),
    'cgg_x3a_encode': (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp]),
    'cgg_x3a_decode': (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp]),
    'cgg_x3_overflow_check': (_c_int, [_c_int, _c_vp, _c_vp]),
    'cgg_gemm_x3s_cfg': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_int] + [_c_int] * 5 + [_c_vp]),
    'cgg_conv_x3s_nhwc_cfg': (_c_int, [_c_vp] * 4 + [_c_int, _c_vp, _c_int] + [_c_int] * 11 + [_c_vp]),
    'cgg_bias_relu_maxpool_nhwc_f32_x3a': (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_group_norm_nhwc_f32_x3a': (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_f, _c_int, _c_vp, _c_int, _c_int, _c_i64, _c_int, _c_vp,
                                             _c_i64, _c_vp, _c_vp, _c_vp]),
    'cgg_encoder_layer_tail_x3a': (_c_int, [_c_vp] * 6 + [_c_f] + [_c_vp] * 6 + [_c_f, _c_vp, _c_int, _c_vp, _c_vp] +
                                   [_c_int] * 3 + [_c_vp]),
    'cgg_wgrad_x3_workspace_bytes': (_c_i64, [_c_int] * 3),
    'cgg_wgrad_x3': (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 3 + [_c_vp]),
    'cgg_wgrad_bias_x3': (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_vp] + [_c_int] * 3 + [_c_vp]),
    'cgg_transpose_f32': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    'cgg_nchw_to_nhwc_pad1_f32': (_c_int, [_c_vp, _c_vp] + [_c_int] * 4 + [_c_vp]),
    'cgg_topk_select': (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    'cgg_absmax_f32': (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    'cgg_relu_bwd_absmax_f32': (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_longlong, _c_vp, _c_vp]),
    'cgg_gemm_x3_bwd': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int, _c_vp] + [_c_int] * 3 + [_c_vp]),
    'cgg_gemm_x3_scaled': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_int] * 4 + [_c_vp]),
    'cgg_conv_x3_nhwc_scaled': (_c_int, [_c_vp] * 6 + [_c_int] * 10 + [_c_vp]),
    'cgg_wgrad_x3_scaled': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp] + [_c_int] * 3 + [_c_vp]),
    'cgg_layernorm_chain': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_int, _c_vp, _c_vp, _c_f, _c_vp, _c_vp,
                                     _c_vp, _c_int, _c_int, _c_int, _c_i64, _c_vp]),
    'cgg_msda_prologue': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_msda_prologue_backward': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_msda_forward_fused_bf16': (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 2 + [_c_int] * 7 + [_c_vp]),
    'cgg_msda_forward_fused_bf16_hm': (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 2 + [_c_int] * 7 + [_c_vp]),
    'cgg_decoder_kv_pack_k': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_decoder_kv_proj_bf16': (_c_int, [_c_vp] * 7 + [_c_int] * 4 + [_c_vp]),
    'cgg_add_layernorm_backward_partials': (ctypes.c_int64, [_c_int]),
    'cgg_add_layernorm_backward': (_c_int, [_c_vp] * 5 + [_c_int, _c_vp, _c_f] + [_c_vp] * 3 + [_c_int, _c_int, _c_vp]),
    'cgg_add_layernorm_backward_amax': (_c_int, [_c_vp] * 5 + [_c_int, _c_vp, _c_f] + [_c_vp] * 4 + [_c_int, _c_int, _c_vp]),
    'cgg_bottleneck64_bf16': (_c_int, [_c_vp] * 8 + [_c_int] * 5 + [_c_vp]),
    'cgg_encoder_proj_pack': (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    'cgg_encoder_proj_bf16': (_c_int, [_c_vp] * 3 + [_c_int] + [_c_vp] * 6 + [_c_int] * 5 + [_c_vp]),
    'cgg_encoder_ffn_ln_kv_bf16': (_c_int, [_c_vp] * 7 + [_c_f, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_vp] * 3 +
                                   [_c_int] * 3 + [_c_vp]),
    'cgg_encoder_layer_tail_bf16': (_c_int, [_c_vp] * 6 + [_c_f] + [_c_vp] * 6 + [_c_f, _c_vp, _c_int, _c_vp, _c_vp, _c_int] +
                                    [_c_vp] * 3 + [_c_int] * 3 + [_c_vp]),
    'cgg_encoder_ffn_ln_bf16': (_c_int, [_c_vp] * 7 + [_c_f, _c_vp, _c_int] + [_c_vp] * 3 + [_c_int] * 3 + [_c_vp]),
    'cgg_add_layernorm_kv': (_c_int, [_c_vp, _c_int, _c_vp, _c_int] + [_c_vp] * 4 + [_c_int, _c_vp, _c_int] + [_c_vp] * 3 +
                             [_c_int, _c_int, _c_f, _c_vp]),
    'cgg_add_layernorm_ex': (_c_int, [_c_vp, _c_int, _c_vp, _c_int] + [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 +
                             [_c_int, _c_int, _c_f, _c_vp]),
    'cgg_group_norm_nhwc_workspace_bytes': (_c_i64, [_c_int] * 3),
    'cgg_group_norm_nhwc_backward_workspace_bytes': (_c_i64, [_c_int] * 3),
    'cgg_group_norm_nhwc_f32_backward': (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [_c_f, _c_int] + [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_group_norm_nhwc_f32_padout': (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_f, _c_int, _c_vp, _c_int, _c_int, _c_i64, _c_int, _c_vp, _c_vp]),
    'cgg_group_norm_nhwc': (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_f, _c_int, _c_vp, _c_int, _c_int, _c_i64, _c_int,
                                       _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_i64, _c_vp]),
    'cgg_group_norm_nhwc_f32': (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_f, _c_int, _c_vp, _c_int, _c_int, _c_i64, _c_int,
                                           _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_i64, _c_vp]),
    'cgg_pack_mask_feature_nhwc_f32_x3': (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    'cgg_point_sample_nhwc_x3': (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_vp]),
    'cgg_pack_mask_feature_nhwc_multi': (_c_int, [_c_vp, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_pack_mask_feature_nhwc': (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_blaslt_init': (_c_int, [ctypes.c_char_p]),
    'cgg_blaslt_set_tuning': (_c_int, [_c_int]),
    'cgg_blaslt_last_tuning': (_c_int, [_c_vp, _c_vp]),
    'cgg_gemm_bias_res_act_bf16': (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    'cgg_im2col3x3_nhwc': (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_stem_conv7x7_packed_bytes': (_c_i64, []),
    'cgg_stem_conv7x7_nchw': (_c_int, [_c_vp] * 3 + [_c_int] * 3 + [_c_vp]),
    'cgg_linear_sum_assignment_f32': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    'cgg_point_sample_planes': (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    'cgg_point_sample_planes_backward': (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    'cgg_point_sample_planes_backward_rows': (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_point_sample_nhwc': (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    'cgg_subsample_nhwc': (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_bias_relu_maxpool_nhwc': (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    'cgg_bias_act_nhwc': (_c_int, [_c_vp] * 3 + [_c_i64, _c_int, _c_int, _c_vp]),
    'cgg_group_norm_workspace_bytes': (_c_i64, [_c_int] * 5),
    'cgg_group_norm': (_c_int, [_c_vp] * 5 + [_c_int] * 5 + [_c_f, _c_int, _c_vp]),
    'cgg_upsample_bilinear': (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    'cgg_instance_masks': (_c_int, [_c_vp] * 6 + [_c_int] * 10 + [_c_vp]),
    'cgg_instance_masks_multi': (_c_int, [_c_vp] * 7 + [_c_int] * 9 + [_c_vp]),
    'cgg_class_topk': (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int] + [_c_vp] * 4),
    'cgg_instance_masks_picks_workspace_bytes': (_c_i64, [_c_int, _c_int]),
    'cgg_instance_masks_picks': (_c_int, [_c_vp] * 3 + [_c_int] + [_c_vp] * 3 + [_c_int] * 10 + [_c_vp]),
    'cgg_panoptic_argmax': (_c_int, [_c_vp] * 6 + [_c_int] * 10 + [_c_vp]),
    'cgg_panoptic_paint': (_c_int, [_c_vp] * 5 + [_c_i64, _c_int, _c_vp]),
    'cgg_rowwise_softmax_argmax': (_c_int, [_c_vp] * 4 + [_c_int, _c_int, _c_vp]),
}

_lib = None


class CggError(RuntimeError):
    """A libcgg_hip.so entry point returned non-zero."""


def load():
    """Load (once) and return the ctypes handle; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CggError(
            f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` '
            '(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for the CGG hot path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cgg_last_error_string()
        raise CggError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (the raw-handle getter is ~10x cheaper than building a
    torch.cuda.Stream object per launch, which showed up as ~1 ms of host time per forward)."""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else device
        if idx is None:
            idx = torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_ptr(t, name='tensor', dtype=None):
    """data_ptr of a contiguous ROCm tensor (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CggError(f'{name} must live on a ROCm device (got {t.device}); the CGG hot path has '
                       'no CPU implementation outside /oracle')
    if not t.is_contiguous():
        raise CggError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise CggError(f'{name} must be {dtype} (got {t.dtype})')
    return ctypes.c_void_p(t.data_ptr())

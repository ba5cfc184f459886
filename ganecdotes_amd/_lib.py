"""ctypes binding of libganecdotes_hip.so (the C ABI declared in include/ganecdotes_hip.h).

There is NO fallback: if the shared object is missing, or an entry point is, importing the
product path fails loudly.  torch is imported first so that the HIP runtime already loaded by
PyTorch-ROCm (SONAME libamdhip64.so.7) is the one the kernels run on.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libganecdotes_hip.so")

c_f = ctypes.c_float
c_i = ctypes.c_int
c_l = ctypes.c_long
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/ganecdotes_hip.h
SIGNATURES = {
    "gx_head_pad": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "gx_head_act_mask": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "gx_head_tap_sum": (c_i, [c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "gx_head_act_mask_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "gx_head_softmax_ce": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gx_head_adam": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_p]),
    "gx_head_argmax": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "gx_confusion": (c_i, [c_p, c_p, c_l, c_i, c_p, c_p]),
    "gx_kmeans_center_panel": (c_i, [c_p, c_i, c_p, c_i, c_l, c_p, c_p, c_p, c_p]),
    "gx_kmeans_assign": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p]),
    "gx_kmeans_accumulate_supported": (c_i, [c_i]),
    "gx_kmeans_accumulate_workspace_bytes": (c_sz, [c_l, c_i, c_i]),
    "gx_kmeans_accumulate": (c_i, [c_p, c_l, c_i, c_l, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "gx_kmeans_update": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "gx_kmeans_row_dist": (c_i, [c_p, c_l, c_i, c_l, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_kmeans_pp_workspace_bytes": (c_sz, [c_i]),
    "gx_kmeans_pp_pots": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gx_kmeans_pp_commit": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_i, c_i, c_p, c_p]),
    "gx_kmeans_pp_sample": (c_i, [c_p, c_l, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "gx_kmeans_label_maps": (c_i, [c_i, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p]),
    "gx_version": (ctypes.c_char_p, []),
    "gx_set_blur_dma": (c_i, [c_i]),
    "gx_set_conv_planes": (c_i, [c_i]),
    "gx_get_conv_planes": (c_i, []),
    "gx_set_conv_tall": (c_i, [c_i]),
    "gx_set_gemm_mode": (c_i, [c_i]),
    "gx_get_gemm_mode": (c_i, []),
    "gx_set_gemm_tall": (c_i, [c_i]),
    "gx_set_gemm_ngroup": (c_i, [c_i]),
    "gx_gemm_f32_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "gx_gemm_f32": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_p, c_p, c_i, c_p,
                          c_sz, c_p]),
    "gx_gemm_f32_cat": (c_i, [c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_p, c_p]),
    "gx_gemm_f32_tn_cat": (c_i, [c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_i, c_p, c_sz, c_p]),
    "gx_gemm_weight_planes_bytes": (c_sz, [c_i, c_i]),
    "gx_split_gemm_weights": (c_i, [c_p, c_i, c_i, c_l, c_p, c_p]),
    "gx_split_gemm_weights_h2": (c_i, [c_p, c_i, c_i, c_l, c_p, c_p]),
    "gx_gemm_f32_nt_planes": (c_i, [c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_l, c_f, c_f, c_p, c_p, c_p, c_i, c_p,
                                    c_p, c_f, c_f, c_p, c_p]),
    "gx_absmax": (c_i, [c_p, c_l, c_i, c_l, c_i, c_p, c_p]),
    "gx_sinkhorn_first_from_parts_workspace_bytes": (c_sz, [c_i, c_i]),
    "gx_sinkhorn_first_from_parts": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_first_preset": (c_i, [c_p, c_p, c_i, c_f, c_p, c_p]),
    "gx_gemm_f32_nt_planes_mlive": (c_i, [c_i, c_i, c_i, c_p, c_l, c_p, c_p, c_l, c_f, c_f, c_p, c_p, c_f, c_p]),
    "gx_gemm_f32_klive": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_p, c_p, c_i, c_p, c_sz, c_p,
                                c_p]),
    "gx_gemm_f32_ranged": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_p, c_p, c_i, c_p, c_sz, c_p,
                                 c_f, c_f, c_p, c_p, c_p]),
    "gx_gemm_f32_tn_cat_ranged": (c_i, [c_i, c_i, c_i, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_f, c_i, c_p, c_sz, c_p,
                                        c_f, c_f, c_p]),
    "gx_gemm_finish_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "gx_gemm_finish_workspace_bytes": (c_sz, [c_l, c_i, c_i]),
    "gx_gemm_f32_nt_planes_finish": (c_i, [c_l, c_i, c_i, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_p, c_sz,
                                           c_p, c_f, c_p, c_p]),
    "gx_gemm_colmax_part_bytes": (c_sz, [c_i, c_i]),
    "gx_fineq_ok": (c_i, [c_i, c_i, c_i]),
    "gx_fineq_pack_weights": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "gx_fineq_ytable_ld": (c_i, []),
    "gx_fineq_fwd": (c_i, [c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_p, c_p]),
    "gx_lowrank_coeff_amax": (c_i, [c_p, c_l, c_i, c_p, c_p]),
    "gx_fineq_bwd_workspace_bytes": (c_sz, []),
    "gx_fineq_bwd": (c_i, [c_p, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p, c_p, c_p,
                           c_p]),
    "gx_proto_raw_tables": (c_i, [c_p, c_l, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_p]),
    "gx_gemm_f32_upadd": (c_i, [c_i, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_p, c_l, c_f, c_p, c_i, c_p]),
    "gx_upsample2_add": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "gx_sumpool2": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "gx_gather_rows_normalize": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_normalize_bwd_scatter": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_proto_small_workspace_bytes": (c_sz, [c_i, c_i]),
    "gx_proto_fwd_small": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_proto_bwd_small": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_sz,
                                 c_p]),
    "gx_lowrank_stride": (c_i, [c_i]),
    "gx_gather_scores": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_proto_fwd_lowrank": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "gx_gather_scores_batch": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "gx_proto_bwd_lowrank_batch": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_sz,
                                         c_p]),
    "gx_swapped_ce_batch": (c_i, [c_p, c_i, c_i, c_l, c_l, c_i, c_f, c_f, c_p, c_f, c_p, c_f, c_f, c_p, c_p, c_sz, c_p]),
    "gx_proto_bwd_lowrank": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_sz, c_p]),
    "gx_proto_lowrank_finish": (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_f, c_p, c_sz, c_p]),
    "gx_upsample2_add_scores": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p]),
    "gx_pixel_norm": (c_i, [c_p, c_l, c_i, c_p, c_p]),
    "gx_equal_linear_skinny": (c_i, [c_p, c_i, c_i, c_p, c_i, c_f, c_p, c_f, c_i, c_p, c_p]),
    "gx_bias_act": (c_i, [c_p, c_l, c_i, c_f, c_p, c_f, c_i, c_p]),
    "gx_lerp_rows": (c_i, [c_p, c_p, c_l, c_i, c_f, c_p, c_p]),
    "gx_view_latents": (c_i, [c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_style_pick": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "gx_demod_batch": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "gx_style_norm": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p]),
    "gx_rotflip_map_batch": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_demod": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "gx_modconv_slab_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gx_conv_weight_planes_bytes": (c_sz, [c_i, c_i, c_i]),
    "gx_split_conv_weights": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_modconv_fwd_p": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_sz,
                               c_p]),
    "gx_modconv_up_convt_p": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gx_modconv_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_sz,
                             c_p]),
    "gx_modconv_up_scratch_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gx_modconv_up_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p,
                                c_p, c_sz, c_p]),
    "gx_modconv_up_convt": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gx_xplanes_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gx_modsplit": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "gx_modconv_fwd_xp": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_p,
                                c_p]),
    "gx_blur_act_fwd_xp": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "gx_modconv_up_convt_xp": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p]),
    "gx_blur_act_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_p]),
    "gx_torgb_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    "gx_nchw_to_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_upfirdn2d_out_shape": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "gx_upfirdn2d": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "gx_fused_bias_act": (c_i, [c_p, c_l, c_p, c_i, c_l, c_p, c_i, c_i, c_f, c_f, c_p, c_p]),
    "gx_random_subset_picks": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_rotflip_map": (c_i, [c_f, c_i, c_i, c_i, c_p, c_p]),
    "gx_gather_taps": (c_i, [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_l, c_p]),
    "gx_stack_norm_map_bilinear": (c_i, [c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_row_sumsq": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p]),
    "gx_norm_map": (c_i, [c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "gx_row_l2_normalize": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p]),
    "gx_l2_normalize_bwd": (c_i, [c_p, c_p, c_p, c_l, c_i, c_l, c_p]),
    "gx_row_argmax": (c_i, [c_p, c_l, c_i, c_l, c_p, c_p]),
    "gx_colsum_workspace_bytes": (c_sz, [c_i]),
    "gx_colsum": (c_i, [c_p, c_l, c_i, c_l, c_f, c_f, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_workspace_bytes": (c_sz, [c_i]),
    "gx_sinkhorn_u": (c_p, [c_p, c_i]),
    "gx_sinkhorn_colmax": (c_i, [c_p, c_l, c_i, c_l, c_p, c_sz, c_p]),
    "gx_sinkhorn_cmax_get": (c_i, [c_p, c_i, c_p, c_p]),
    "gx_sinkhorn_cmax_set": (c_i, [c_p, c_i, c_p, c_p]),
    "gx_sinkhorn_pass": (c_i, [c_p, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_f, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_update": (c_i, [c_p, c_p, c_i, c_i, c_p, c_f, c_p, c_p]),
    "gx_sinkhorn": (c_i, [c_p, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_small": (c_i, [c_p, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_small_batch": (c_i, [c_p, c_i, c_l, c_l, c_i, c_l, c_f, c_i, c_p, c_l, c_p, c_l, c_p, c_p, c_sz,
                                      c_p]),
    "gx_sinkhorn_small_batch_colmax": (c_i, [c_p, c_i, c_l, c_l, c_i, c_l, c_p, c_sz, c_p, c_p]),
    "gx_sinkhorn_small_batch_pass": (c_i, [c_p, c_i, c_l, c_l, c_i, c_l, c_f, c_i, c_p, c_l, c_p, c_l, c_f, c_p, c_p, c_p,
                                           c_sz, c_p, c_p]),
    "gx_sinkhorn_small_batch_finish": (c_i, [c_i, c_i, c_f, c_i, c_p, c_l, c_p, c_p, c_p, c_p]),
    "gx_sinkhorn_codes": (c_i, [c_p, c_l, c_i, c_l, c_f, c_p, c_p, c_l, c_p]),
    "gx_hist_marginal": (c_i, [c_p, c_l, c_i, c_p, c_p, c_i, c_p]),
    "gx_hist_range": (c_i, [c_p, c_l, c_p, c_p, c_p]),
    "gx_hist_counts": (c_i, [c_p, c_l, c_p, c_i, c_p, c_p, c_p]),
    "gx_hist_finish": (c_i, [c_p, c_i, c_p]),
    "gx_sinkhorn_bigk_batch_workspace_bytes": (c_sz, [c_i, c_l, c_i]),
    "gx_sinkhorn_bigk_batch": (c_i, [c_p, c_i, c_l, c_l, c_i, c_l, c_f, c_i, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_table_ok": (c_i, [c_i, c_i]),
    "gx_sinkhorn_table_workspace_bytes": (c_sz, [c_i, c_l, c_i, c_i]),
    "gx_sinkhorn_table_first": (c_i, [c_p, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_table_sweep": (c_i, [c_p, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_l, c_f, c_p, c_p, c_sz,
                                      c_p]),
    "gx_sinkhorn_table_update": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_f, c_p, c_p, c_p]),
    "gx_sinkhorn_table_rebase": (c_i, [c_p, c_p, c_p, c_l, c_f, c_p]),
    "gx_sinkhorn_table": (c_i, [c_p, c_p, c_l, c_i, c_p, c_l, c_i, c_p, c_p, c_i, c_f, c_i, c_p, c_p, c_l, c_f, c_p, c_p, c_p, c_p, c_p,
                                c_sz, c_p]),
    "gx_sinkhorn_rows_batch_workspace_bytes": (c_sz, [c_i, c_l, c_i]),
    "gx_sinkhorn_rows_batch": (c_i, [c_p, c_p, c_i, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gx_swapped_ce_workspace_bytes": (c_sz, [c_l, c_i]),
    "gx_swapped_ce_fused_workspace_bytes": (c_sz, [c_i]),
    "gx_swapped_ce_fused": (c_i, [c_p, c_p, c_l, c_i, c_l, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_l, c_f, c_f, c_p, c_p, c_f,
                                  c_p, c_sz, c_p]),
    "gx_swapped_ce_pixels_tab": (c_i, [c_p, c_p, c_i, c_i, c_l, c_p, c_l, c_i, c_f, c_f, c_p, c_f, c_p, c_p, c_l, c_i, c_f, c_f,
                                       c_p, c_p, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_colmax_rows": (c_i, [c_p, c_p, c_l, c_i, c_l, c_p, c_sz, c_p]),
    "gx_sinkhorn_pass_rows": (c_i, [c_p, c_p, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_f, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_first_rows_workspace_bytes": (c_sz, [c_i]),
    "gx_sinkhorn_first_rows": (c_i, [c_p, c_p, c_l, c_i, c_l, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "gx_sinkhorn_rows": (c_i, [c_p, c_p, c_l, c_i, c_l, c_f, c_i, c_p, c_p, c_f, c_p, c_p, c_sz, c_p]),
    "gx_swapped_ce_pixels": (c_i, [c_p, c_i, c_l, c_i, c_l, c_p, c_l, c_i, c_f, c_f, c_p, c_f, c_p, c_p, c_l, c_i, c_f, c_f, c_p,
                                   c_p, c_f, c_p, c_sz, c_p]),
    "gx_swapped_ce": (c_i, [c_p, c_p, c_l, c_i, c_l, c_f, c_f, c_p, c_p, c_f, c_p, c_p, c_l, c_f, c_f, c_p, c_p,
                            c_sz, c_p]),
    "gx_gather_rows": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "gx_scatter_rows_add": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "gx_leaky_relu": (c_i, [c_p, c_l, c_f, c_p]),
    "gx_leaky_relu_bwd": (c_i, [c_p, c_p, c_l, c_f, c_p]),
    "gx_col_moment_workspace_bytes": (c_sz, [c_i]),
    "gx_col_moment": (c_i, [c_p, c_l, c_i, c_p, c_i, c_f, c_p, c_p, c_sz, c_p]),
    "gx_bn_act_fwd": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_f, c_i, c_f, c_p, c_p, c_p]),
    "gx_bn_act_bwd_sums": (c_i, [c_p, c_p, c_l, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "gx_bn_act_bwd_apply": (c_i, [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_f, c_i, c_f, c_p, c_p, c_f, c_p, c_p]),
    "gx_col_stats_workspace_bytes": (c_sz, [c_i]),
    "gx_col_stats": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gx_bn_lrelu_fwd": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p]),
    "gx_bn_lrelu_bwd": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p]),
    "gx_simclr_ntxent": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "gx_row_div": (c_i, [c_p, c_l, c_i, c_p, c_p]),
    "gx_sumsq_workspace_bytes": (c_sz, []),
    "gx_sumsq": (c_i, [c_p, c_l, c_p, c_p, c_sz, c_p]),
    "gx_larc_sgd": (c_i, [c_p, c_p, c_p, c_l, c_p, c_f, c_f, c_f, c_i, c_i, c_p]),
}

ERRORS = {-1: "bad argument", -2: "misaligned pointer / leading dimension", -3: "workspace too small",
          -4: "kernel launch failed (hipGetLastError)"}

_lib = None


def load():
    """dlopen the library and type every entry point.  Raises RuntimeError when the library or
    a symbol is missing (build it with `python -m ganecdotes_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    lab = os.environ.get("GX_LAB_LIB")
    if lab:
        # a lab build (tools/lab/labbuild.sh: ONE translation unit recompiled with ablation flags, some of which compute wrong
        # values on purpose) lives in its own file and is only ever loaded on request — and says so
        import sys
        print(f"[ganecdotes_amd] GX_LAB_LIB set: loading the LAB library {lab} instead of {LIB_PATH}", file=sys.stderr, flush=True)
        path = lab
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is required (no CPU/PyTorch fallback). "
            "Build it with `python -m ganecdotes_amd.build` (needs hipcc).")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libganecdotes_hip.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)} (code {rc})")

"""ctypes binding of libjg355.so (the C ABI declared in include/jg355.h).

The library is built in-tree with hipcc for gfx950 (`build()`), loaded once, and every entry
point gets its argtypes from the table below.  There is NO fallback: if the shared object is
missing or a symbol is absent, importing the ops fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libjg355.so")
SOURCES = ["attention.hip", "gemm_nt.hip", "conv_halo.hip", "conv_kxk.hip", "reflect_border.hip", "conv_p64.hip", "conv1x1.hip", "gemm_tn.hip", "wgrad_halo.hip", "wgrad_sw.hip", "wgrad_kxk.hip", "nce.hip", "segformer.hip", "vit.hip", "projected_d.hip", "effnet.hip", "norm.hip", "gn_fused.hip", "elementwise.hip", "optim.hip", "capi.hip"]
# -fno-slp-vectorize: the SLP vectoriser packs independent fp32 chains into v_pk_* pairs (register tuples: gn_fused.hip went from 150 spilled
# registers to none without it) -- "an anti-lever beside MFMAs" in the MI355X guide; same-box A/B of the whole step: 52.2 -> 52.0 ms
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize"]

FILE_FLAGS = {}      # per-source extra flags

JG_OK, JG_ERR_BAD_ARG, JG_ERR_UNSUPPORTED, JG_ERR_LAUNCH = 0, -1, -2, -3
JG_F16, JG_BF16 = 0, 1
JG_ACT_NONE, JG_ACT_SILU, JG_ACT_RELU, JG_ACT_LRELU, JG_ACT_TANH = 0, 1, 2, 3, 4
JG_OUT_ATOMIC_F32, JG_OUT_STORE_F32, JG_OUT_STORE_T = 0, 1, 2

c_i32, c_i64, c_f32, c_p = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    _fields_ = (
        [(n, c_p) for n in ("x", "w", "y", "bias", "res")]
        + [(n, c_i32) for n in ("B", "H", "W", "Cin", "Cout", "R", "S", "pad", "stride", "Ho", "Wo")]
        + [(n, c_i64) for n in ("ldx", "ldw", "ldy", "ldres")]
        + [(n, c_i32) for n in ("nbatch", "nh")]
        + [(n, c_i64) for n in ("sxb", "sxh", "swb", "swh", "syb", "syh", "srb", "srh")]
        + [("alpha", c_f32), ("res_scale", c_f32), ("out_f32", c_i32), ("stats", c_p), ("ldstats", c_i64), ("stats_slots", c_i32), ("stats_mode", c_i32), ("gn_x", c_p), ("gn_ldx", c_i64), ("gn_ab", c_p), ("gn_act", c_i32), ("pad_mode", c_i32), ("res_mode", c_i32), ("x_mode", c_i32), ("y_mode", c_i32), ("ws", c_p), ("ws_bytes", c_i64)]
    )


class WgradArgs(C.Structure):
    _fields_ = (
        [(n, c_p) for n in ("dy", "x", "dw", "dbias")]
        + [(n, c_i32) for n in ("B", "H", "W", "Cin", "Cout", "R", "S", "pad", "stride", "Ho", "Wo")]
        + [(n, c_i32) for n in ("Cin_out", "Cout_out")]
        + [(n, c_i64) for n in ("lddy", "ldx", "lddw")]
        + [(n, c_i32) for n in ("nbatch", "nh", "splitk")]
        + [(n, c_i64) for n in ("sdyb", "sdyh", "sxb", "sxh", "sdwb", "sdwh")]
        + [("alpha", c_f32), ("out_mode", c_i32), ("dbias_scale", c_f32), ("pad_mode", c_i32), ("x_mode", c_i32)]
    )


# name -> argtypes (restype is int except where noted); mirrors include/jg355.h one to one
SIGNATURES = {
    "jg_version": [],
    "jg_strerror": [c_i32],
    "jg_set_tuning": [C.c_char_p, c_i32],
    "jg_get_tuning": [C.c_char_p],
    "jg_last_kernel": [],
    "jg_conv2d_nt": [c_i32, C.POINTER(ConvArgs), c_p],
    "jg_conv1x1_gn_apply": [c_i32, C.POINTER(ConvArgs), c_p, c_p, c_i64, c_i32, c_p],
    "jg_conv1x1_gn_bwd_apply": [c_i32, C.POINTER(ConvArgs), c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_f32, c_p, c_i64, c_f32, c_i32, c_p],
    "jg_conv2d_wgrad_tn": [c_i32, C.POINTER(WgradArgs), c_p],
    "jg_conv2d_wgrad_tn_group": [c_i32, C.POINTER(WgradArgs), c_i32, c_p],
    "jg_gn_stats": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_gn_coef": [c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_gn_apply": [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_apply_add": [c_i32, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_reduce": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_subpixel_fold": [c_i32, c_p, c_p, c_i32, c_i32, c_p],
    "jg_transposed_fold": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_apply_pool": [c_i32, c_p, c_i64, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_gn_bwd_reduce_up": [c_i32, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_reduce_up_acc": [c_i32, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_fused": [c_i32, c_i32, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p, c_i64,
                        c_p, c_i64, c_f32, c_p, c_i64, c_f32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_apply_fc": [c_i32, c_i32, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p, c_i64,
                           c_p, c_i64, c_f32, c_p, c_i64, c_f32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_reduce_ld_acc": [c_i32, c_p, c_i64, c_p, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_apply_up": [c_i32, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_p, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_i64, c_f32, c_i32, c_i32,
                           c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_coef": [c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_coef_slots": [c_p, c_i32, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_apply": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_stats_ld": [c_i32, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_p],
    "jg_gn_coef_ld": [c_p, c_i64, c_i32, c_p, c_p, c_p, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_gn_apply_ld": [c_i32, c_p, c_i64, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_reduce_ld": [c_i32, c_p, c_i64, c_p, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gn_bwd_apply_ld": [c_i32, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_f32, c_p, c_i64, c_f32,
                           c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_pool2x2": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_upsample2x": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_pool2x2_ld": [c_i32, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_upsample2x_ld": [c_i32, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_copy_channels": [c_i32, c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, c_i32, c_p],
    "jg_axpby": [c_i32, c_p, c_f32, c_p, c_p, c_f32, c_p, c_i64, c_p],
    "jg_transpose_heads": [c_i32, c_p, c_i64, c_i64, c_i64, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_softmax_fwd": [c_i32, c_p, c_p, c_i64, c_i32, c_p],
    "jg_softmax_bwd": [c_i32, c_p, c_p, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_attention_fwd": [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_attention_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_linear_fwd": [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_linear_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_gamma_embedding": [c_p, c_p, c_i32, c_i32, c_f32, c_p],
    "jg_ddpm_prepare": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_ddpm_mse_loss": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_p],
    "jg_act_fwd": [c_i32, c_p, c_p, c_i64, c_i32, c_p],
    "jg_act_bwd": [c_i32, c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_reflect_pad2d": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_crop2d": [c_i32, c_p, c_p] + [c_i32] * 9 + [c_p],
    "jg_reflect_pad2d_bwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_reflect_dgrad_border": [c_i32, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_reflect_dgrad_border_ws_floats": [c_i32, c_i32, c_i32, c_i32],
    "jg_dilate2d": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_tapsum7": [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_tapspread7": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_conv_dgrad_gather": [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_subsample2d": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_channel_sum": [c_i32, c_p, c_i64, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_ddpm_multiscale_loss": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32,
                                c_f32, c_p],
    "jg_gather_rows": [c_i32, c_p, c_i64, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_p],
    "jg_scatter_rows": [c_i32, c_p, c_i64, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_p],
    "jg_gather_rows_grouped": [c_i32, c_p, c_i64, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_scatter_rows_grouped": [c_i32, c_p, c_i64, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_l2norm_fwd": [c_p, c_p, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_l2norm_bwd": [c_p, c_p, c_p, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_lsgan_loss": [c_i32, c_p, c_f32, c_p, c_p, c_i64, c_i32, c_f32, c_f32, c_p],
    "jg_gan_loss": [c_i32, c_i32, c_p, c_f32, c_p, c_p, c_i64, c_i32, c_f32, c_f32, c_p],
    "jg_sgemm": [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_i64, c_i64,
                 c_i64, c_f32, c_f32, c_i32, c_i32, c_i32, c_p],
    "jg_row_axpy": [c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_nce_sinkhorn_fwd": [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_nce_ce": [c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i32, c_i32, c_f32, c_f32, c_p, c_p, c_f32, c_p],
    "jg_nce_sinkhorn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_layernorm_fwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_vit_attention_fwd": [c_i32, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_vit_attention_bwd": [c_i32, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_vit_tokens_fwd": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_vit_tokens_bwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_gelu_fwd": [c_i32, c_p, c_p, c_i64, c_p],
    "jg_gelu_bwd": [c_i32, c_p, c_p, c_p, c_i64, c_p],
    "jg_layernorm_bwd_res": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_unpatchify": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_transpose2d": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_layernorm_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_layernorm_bwd_add": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_layernorm_fwd_add": [c_i32, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_f32, c_p],
    "jg_layernorm_bwd_add2": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i32, c_p],
    "jg_dwconv3x3_fwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_dwconv3x3_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_dwconv3x3_bwd_ws": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_dwconv3x3_bwd_ws_floats": [c_i32, c_i32, c_i32, c_i32],
    "jg_dwconv3x3_fwd_pad": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_dwconv3x3_bwd_ws_pad": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_attn_smallkv_fwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_f32, c_p],
    "jg_attn_smallkv_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64,
                            c_f32, c_p],
    "jg_attn_smallkv_bwd2": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64,
                             c_f32, c_p],
    "jg_bilinear_fwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_p],
    "jg_bilinear_bwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_p],
    "jg_resize_sum_bwd_ws_floats": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "jg_resize_sum_bwd": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p],
    "jg_resize_sum": [c_i32, c_p, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p],
    "jg_bilinear2_fwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_p],
    "jg_bilinear2_bwd": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_p],
    "jg_spectral_power_iter": [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_f32, c_p],
    "jg_spectral_weights": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_spectral_wgrad_fix": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p],
    "jg_spectral_group_forward": [c_i32, c_p, c_i32, c_p, c_i64, c_p, c_i32, c_i32, c_i64, c_f32, c_p],
    "jg_spectral_group_wgrad_fix": [c_p, c_i32, c_p, c_p, c_p, C.c_uint64, c_i64, c_p],
    "jg_dwconv_affine_act_fwd": [c_i32, c_p, c_p, c_p, c_p, c_p] + [c_i32] * 11 + [c_p],
    "jg_dwconv_affine_act_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p] + [c_i32] * 11 + [c_p],
    "jg_chan_affine_act_fwd": [c_i32, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p],
    "jg_chan_affine_act_bwd": [c_i32, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p],
    "jg_hinge_loss": [c_i32, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_f32, c_f32, c_p],
    "jg_bn_coef": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_p],
    "jg_bn_bwd_coef": [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_attn_compose_fwd": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_attn_compose_bwd": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_scale": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_p],
    "jg_ddpm_p_sample": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_cm_noisy": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_cm_combine": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_cm_loss": [c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32,
                   c_f32, c_f32, c_f32, c_p],
    "jg_noise_level_embedding": [c_p, c_p, c_p, c_i32, c_i32, c_p],
    "jg_noise_level_embedding_bwd": [c_p, c_p, c_p, c_p, c_i32, c_i32, c_p],
    "jg_resample_u8": [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_resize_nearest_u8": [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_input_pipeline": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_nhwc_to_nchw_f32": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_nchw_f32_to_nhwc": [c_i32, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p],
    "jg_adamw_ema": [c_p, c_p, c_p, c_p, c_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_f32, c_f32, c_i32, c_p],
    "jg_optim_step": [c_i32, c_p, c_p, c_p, c_p, c_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_f32, c_i32, c_p, c_p, c_p],
    "jg_adamw_ema_skip": [c_p, c_p, c_p, c_p, c_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_f32, c_f32, c_i32, c_p, c_p, c_p],
    "jg_grad_nonfinite": [c_p, c_i64, c_p, c_p],
    "jg_ema_update": [c_p, c_p, c_i64, c_f32, c_p],
    "jg_refresh_weights": [c_i32, c_p, c_p, c_p, c_p, c_i32, c_p],
}


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU) into csrc/libjg355.so.
    One object per source under csrc/build/ (compiled in parallel, rebuilt when the source or any header is newer), then one link.
    csrc/build/BUILD_INFO.json records what was compiled by this call."""
    import json
    import time
    from concurrent.futures import ThreadPoolExecutor

    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(os.path.dirname(_HERE), "include", "jg355.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    todo, objs = [], []
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            todo.append((sp, op))

    def cc(job):
        sp, op = job
        cmd = [hipcc] + cflags + FILE_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    t0 = time.time()
    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(cc, todo))
    relink = bool(todo) or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs)
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    if todo or relink:
        with open(os.path.join(bdir, "BUILD_INFO.json"), "w") as f:
            json.dump({"compiled": [os.path.basename(s) for s, _ in todo], "linked": relink, "seconds": round(time.time() - t0, 1),
                       "flags": cflags, "hipcc": hipcc}, f)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library with typed entry points.  Raises if it is missing -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  joligen_amd has no CPU / eager fallback."
        )
    L = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is absent
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name in ("jg_strerror", "jg_last_kernel") else c_i64 if name.endswith("_ws_floats") else c_i32
    _lib = L
    return L


def set_tuning(name: str, value: int):
    """override a dispatch switch of the library for the rest of the process; returns the previous value"""
    L = lib()
    prev = L.jg_get_tuning(name.encode())
    check(L.jg_set_tuning(name.encode(), int(value)), f"jg_set_tuning({name})")
    return prev


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().jg_strerror(code).decode()
        raise RuntimeError(f"libjg355 {what} failed: {msg} ({code})")

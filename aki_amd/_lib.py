"""ctypes binding of libaki_mi355x.so (include/aki_mi355x.h).  The product path has no CPU fallback:
if the library is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# AKI_MI355X_LIB: A/B hook for lab builds of the same ABI (tools/attn_*.py); the product path never sets it
LIB_PATH = os.environ.get("AKI_MI355X_LIB") or os.path.join(_HERE, "lib", "libaki_mi355x.so")
LAB_LIB_PATH = os.path.join(_HERE, "lib", "libaki_mi355x_lab.so")

AKI_DT_BF16, AKI_DT_F32, AKI_DT_FP8_E4M3, AKI_DT_W8A16 = 0, 1, 2, 3
AKI_ACT_NONE, AKI_ACT_GELU_ERF, AKI_ACT_GELU_TANH, AKI_ACT_SWIGLU = 0, 1, 2, 3
AKI_DEAD_ROWS_ZERO, AKI_DEAD_ROWS_UNIFORM = 0, 1
AKI_MAX_RECTS = 8
AKI_PLAN_STRIDE = 12
AKI_ABI_VERSION = 17


class AkiError(RuntimeError):
    pass


class MmaRect(C.Structure):
    _fields_ = [("row_lo", C.c_int32), ("row_hi", C.c_int32), ("col_lo", C.c_int32), ("col_hi", C.c_int32)]


class MmaAttnCoreArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("rects", C.c_void_p), ("col_valid_bits", C.c_void_p), ("seq_lens", C.c_void_p),
                ("max_rects", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("Dh", C.c_int32),
                ("scale", C.c_float), ("dtype", C.c_int32), ("dead_rows", C.c_int32), ("kv_capacity", C.c_int32)]


class MmaAttnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w_qkv", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p),
                ("position_ids", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p), ("rects", C.c_void_p),
                ("col_valid_bits", C.c_void_p), ("seq_lens", C.c_void_p), ("max_rects", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("Dh", C.c_int32), ("d_model", C.c_int32),
                ("ldx", C.c_int32), ("ldw", C.c_int32), ("pos_rows", C.c_int32), ("scale", C.c_float),
                ("dtype", C.c_int32), ("dead_rows", C.c_int32), ("kv_capacity", C.c_int32),
                ("x_scale", C.c_void_p), ("w_scale", C.c_void_p), ("row_scale", C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("q_stride_b", C.c_int64), ("q_stride_h", C.c_int64), ("q_stride_t", C.c_int64),
                ("k_stride_b", C.c_int64), ("k_stride_h", C.c_int64), ("k_stride_t", C.c_int64),
                ("v_stride_b", C.c_int64), ("v_stride_h", C.c_int64), ("v_stride_t", C.c_int64),
                ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("Dh", C.c_int32),
                ("scale", C.c_float), ("dtype", C.c_int32), ("lse", C.c_void_p)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("d_o", C.c_void_p), ("lse", C.c_void_p),
                ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("rects", C.c_void_p), ("max_rects", C.c_int32),
                ("col_valid_bits", C.c_void_p), ("seq_lens", C.c_void_p), ("masked", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("Dh", C.c_int32),
                ("scale", C.c_float), ("dtype", C.c_int32)]


class LinearArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldx", C.c_int32), ("ldw", C.c_int32),
                ("ldy", C.c_int32), ("ldr", C.c_int32), ("res_row_mod", C.c_int32), ("act", C.c_int32),
                ("dtype", C.c_int32), ("x_scale", C.c_void_p), ("w_scale", C.c_void_p),
                ("w2", C.c_void_p), ("w2_row0", C.c_int32), ("w2_rows", C.c_int32),
                ("row_scale", C.c_void_p), ("row_shift", C.c_void_p), ("col_shift", C.c_void_p), ("stats_rstd", C.c_void_p),
                ("stats_mean", C.c_void_p), ("stats_eps", C.c_float), ("stats_workspace", C.c_void_p),
                ("stats_workspace_bytes", C.c_size_t), ("splitk_workspace", C.c_void_p), ("splitk_workspace_bytes", C.c_size_t),
                ("preact_out", C.c_void_p), ("ld_preact", C.c_int64)]


class DecoderLayer(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("w_o", C.c_void_p), ("w_gate_up", C.c_void_p), ("w_down", C.c_void_p), ("k_cache", C.c_void_p),
                ("v_cache", C.c_void_p)]


class DecoderStackArgs(C.Structure):
    _fields_ = [("layers", C.POINTER(DecoderLayer)), ("n_layers", C.c_int32), ("h_in", C.c_void_p), ("h_out", C.c_void_p), ("rstd_out", C.c_void_p),
                ("cos", C.c_void_p), ("sin", C.c_void_p), ("position_ids", C.c_void_p), ("pos_rows", C.c_int32), ("rects", C.c_void_p),
                ("col_valid_bits", C.c_void_p), ("seq_lens", C.c_void_p), ("max_rects", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
                ("L", C.c_int32), ("Dh", C.c_int32), ("d", C.c_int32), ("F", C.c_int32), ("kv_capacity", C.c_int32), ("scale", C.c_float),
                ("rms_eps", C.c_float), ("dead_rows", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("stats_workspace", C.c_void_p), ("stats_workspace_bytes", C.c_size_t), ("splitk_workspace", C.c_void_p),
                ("splitk_workspace_bytes", C.c_size_t)]


class SiglipLayer(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("c_qkv", C.c_void_p), ("w_out", C.c_void_p), ("b_out", C.c_void_p),
                ("w_fc1", C.c_void_p), ("b_fc1", C.c_void_p), ("c_fc1", C.c_void_p), ("w_fc2", C.c_void_p), ("b_fc2", C.c_void_p)]


class SiglipStackArgs(C.Structure):
    _fields_ = [("layers", C.POINTER(SiglipLayer)), ("n_layers", C.c_int32), ("h_in", C.c_void_p), ("h_out", C.c_void_p), ("fc1_out", C.c_void_p),
                ("N", C.c_int32), ("L", C.c_int32), ("E", C.c_int32), ("heads", C.c_int32), ("I", C.c_int32), ("Ip", C.c_int32),
                ("act", C.c_int32), ("ln_eps", C.c_float), ("scale", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("stats_workspace", C.c_void_p), ("stats_workspace_bytes", C.c_size_t), ("splitk_workspace", C.c_void_p),
                ("splitk_workspace_bytes", C.c_size_t)]


class PerceiverLayer(C.Structure):
    _fields_ = [("norm_media_w", C.c_void_p), ("norm_media_b", C.c_void_p), ("norm_latents_w", C.c_void_p), ("norm_latents_b", C.c_void_p),
                ("w_q", C.c_void_p), ("w_kv", C.c_void_p), ("w_out", C.c_void_p), ("ff_ln_w", C.c_void_p), ("ff_ln_b", C.c_void_p),
                ("ff_w1", C.c_void_p), ("ff_w2", C.c_void_p)]


class PerceiverStackArgs(C.Structure):
    _fields_ = [("layers", C.POINTER(PerceiverLayer)), ("n_layers", C.c_int32), ("x", C.c_void_p), ("latents", C.c_void_p), ("norm_w", C.c_void_p),
                ("norm_b", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("out", C.c_void_p), ("n1", C.c_int32), ("n2", C.c_int32),
                ("D", C.c_int32), ("heads", C.c_int32), ("dim_head", C.c_int32), ("d_ff", C.c_int32), ("D_out", C.c_int32), ("scale", C.c_float),
                ("eps", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class SpliceArgs(C.Structure):
    _fields_ = [("lang_x", C.c_void_p), ("attention_mask", C.c_void_p), ("labels", C.c_void_p),
                ("embed_weight", C.c_void_p), ("embed_additional", C.c_void_p), ("vision_tokens", C.c_void_p),
                ("plan", C.c_void_p), ("inputs_embeds", C.c_void_p), ("labels_out", C.c_void_p),
                ("mask_1d_out", C.c_void_p), ("rects", C.c_void_p), ("col_valid_bits", C.c_void_p),
                ("seq_lens", C.c_void_p), ("max_original_id", C.c_int64), ("media_token_id", C.c_int64),
                ("pad_token_id", C.c_int64), ("B", C.c_int32), ("T", C.c_int32), ("T_img", C.c_int32),
                ("Nv", C.c_int32), ("d", C.c_int32), ("L_out", C.c_int32), ("max_rects", C.c_int32),
                ("padding_side", C.c_int32), ("dtype", C.c_int32)]


class DecodeChainLayer(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("w_o", C.c_void_p), ("w_gate_up", C.c_void_p), ("w_down", C.c_void_p), ("norm1", C.c_void_p),
                ("norm2", C.c_void_p), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("s_qkv", C.c_void_p), ("s_o", C.c_void_p),
                ("s_gate_up", C.c_void_p), ("s_down", C.c_void_p)]


class DecodeChainArgs(C.Structure):
    _fields_ = [("layers", C.c_void_p), ("h_in", C.c_void_p), ("h_out", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p),
                ("cache_len", C.c_void_p), ("col_valid_bits", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("n_layers", C.c_int32), ("nwords", C.c_int32), ("d", C.c_int32), ("H", C.c_int32), ("Dh", C.c_int32), ("F", C.c_int32),
                ("capacity", C.c_int32), ("max_keys", C.c_int32), ("scale", C.c_float), ("rms_eps", C.c_float), ("dtype", C.c_int32),
                ("batch", C.c_int32)]


# name -> (restype, argtypes); also the list of symbols include/aki_mi355x.h declares
SIGNATURES = {
    "aki_strerror": (C.c_char_p, [C.c_int]),
    "aki_abi_version": (C.c_int, []),
    "aki_mma_attn_core_workspace_bytes": (C.c_size_t, [C.c_int32] * 5),
    "aki_mma_attn_core_fwd": (C.c_int, [C.POINTER(MmaAttnCoreArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_mma_attn_workspace_bytes": (C.c_size_t, [C.c_int32] * 5),
    "aki_mma_attn_fwd": (C.c_int, [C.POINTER(MmaAttnArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_qkv_rope_fwd": (C.c_int, [C.POINTER(MmaAttnArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_linear_fwd": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    "aki_linear_stats_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "aki_linear_splitk_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "aki_linear_stats_counter_bytes": (C.c_size_t, [C.c_int32]),
    "aki_row_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "aki_rmsnorm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_float, C.c_int32, C.c_void_p]),
    "aki_layernorm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "aki_patch_embed_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "aki_patch_embed_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_connector_mlp_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "aki_connector_mlp_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_void_p, C.c_size_t,
                                                                            C.c_void_p]),
    "aki_connector_proj_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_void_p, C.c_size_t,
                                                                             C.c_void_p]),
    "aki_rope_append_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 5 + [C.c_void_p]),
    "aki_decode_attn_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "aki_decode_attn_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 6 + [C.c_float, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_decode_attn_fused_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 6 + [C.c_float, C.c_int32, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "aki_decode_linear_fwd": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p, C.c_float, C.c_void_p]),
    "aki_decode_chain_workspace_bytes": (C.c_size_t, [C.c_int32] * 5),
    "aki_decode_chain_error_offset": (C.c_size_t, [C.c_int32] * 2),
    "aki_decode_chain_batch_workspace_bytes": (C.c_size_t, [C.c_int32] * 6),
    "aki_decode_chain_batch_error_offset": (C.c_size_t, [C.c_int32] * 3),
    "aki_decode_chain_fwd": (C.c_int, [C.POINTER(DecodeChainArgs), C.c_void_p]),
    "aki_attn_bwd_workspace_bytes": (C.c_size_t, [C.c_int32] * 3),
    "aki_attn_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_transpose": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]),
    "aki_gemm_tn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 3 + [C.c_int64] * 3 + [C.c_int32, C.c_void_p]),
    "aki_norm_bwd_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "aki_norm_bwd": (C.c_int, [C.c_int32] + [C.c_void_p] * 7 + [C.c_int32] * 6 + [C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                                                                  C.c_void_p]),
    "aki_colsum_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "aki_colsum": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_swiglu_fwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]),
    "aki_swiglu_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 6 + [C.c_void_p]),
    "aki_gelu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "aki_gelu_bwd": (C.c_int, [C.c_void_p] * 3 + [C.c_size_t, C.c_int32, C.c_void_p]),
    "aki_rope_bwd_merge": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p]),
    "aki_ce_loss_fwd_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_float, C.c_int32, C.c_void_p]),
    "aki_ce_rows_fwd_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_float, C.c_int32, C.c_void_p]),
    "aki_grad_sqnorm_workspace_bytes": (C.c_size_t, []),
    "aki_grad_sqnorm": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aki_adamw_step": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p] + [C.c_float] * 7 + [C.c_int32, C.c_void_p]),
    "aki_adamw_step_t": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p] + [C.c_float] * 7 + [C.c_int32, C.c_int32, C.c_void_p]),
    "aki_adamw_step_g32": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p] + [C.c_float] * 7 + [C.c_int32, C.c_void_p]),
    "aki_quant_rows_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "aki_splice_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "aki_splice_fwd": (C.c_int, [C.POINTER(SpliceArgs), C.c_void_p]),
    "aki_mma_mask_dense": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "aki_greedy_pick": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "aki_decoder_stack_workspace_bytes": (C.c_size_t, [C.c_int32] * 7),
    "aki_decoder_stack_fwd": (C.c_int, [C.POINTER(DecoderStackArgs), C.c_void_p]),
    "aki_siglip_stack_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "aki_siglip_stack_fwd": (C.c_int, [C.POINTER(SiglipStackArgs), C.c_void_p]),
    "aki_perceiver_stack_workspace_bytes": (C.c_size_t, [C.c_int32] * 6),
    "aki_perceiver_stack_fwd": (C.c_int, [C.POINTER(PerceiverStackArgs), C.c_void_p]),
    "aki_greedy_pick_embed": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                        C.c_void_p]),
    "aki_sft_collate_pad": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32] + [C.c_void_p] * 4),
    "aki_mma_mask_to_table_workspace_bytes": (C.c_size_t, [C.c_int32] * 2),
    "aki_mma_mask_to_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def _bind(path: str) -> C.CDLL:
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AkiError(f"{path} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.aki_abi_version() != AKI_ABI_VERSION:
        raise AkiError(f"{path}: ABI version mismatch; rebuild it")
    return lib


def load_lab() -> C.CDLL:
    """The lab twin of the library (same ABI + aki_lab_set_gemm_tile).  Tests / tools only: `with use_lab(mode): ...`."""
    import torch  # noqa: F401
    if not os.path.exists(LAB_LIB_PATH):
        raise AkiError(f"{LAB_LIB_PATH} not found: build it with `python -m aki_amd.build`")
    lib = _bind(LAB_LIB_PATH)
    lib.aki_lab_set_gemm_tile.restype = None
    lib.aki_lab_set_gemm_tile.argtypes = [C.c_int]
    lib.aki_lab_set_attn_variant.restype = None
    lib.aki_lab_set_attn_variant.argtypes = [C.c_int]
    lib.aki_lab_set_probe_block.restype = None
    lib.aki_lab_set_probe_block.argtypes = [C.c_int]
    lib.aki_lab_set_clock_probe.restype = None
    lib.aki_lab_set_clock_probe.argtypes = [C.c_void_p]
    lib.aki_lab_set_chain.restype = None
    lib.aki_lab_set_chain.argtypes = [C.c_int] * 4
    lib.aki_lab_set_chain_nb.restype = None
    lib.aki_lab_set_chain_nb.argtypes = [C.c_int]
    lib.aki_lab_set_chain_touch.restype = None
    lib.aki_lab_set_chain_touch.argtypes = [C.c_int]
    lib.aki_lab_set_chain_stamps.restype = None
    lib.aki_lab_set_chain_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.aki_lab_set_chain_lds.restype = None
    lib.aki_lab_set_chain_lds.argtypes = [C.c_int]
    lib.aki_lab_set_slice_major.restype = None
    lib.aki_lab_set_slice_major.argtypes = [C.c_int]
    lib.aki_lab_set_small_m.restype = None
    lib.aki_lab_set_small_m.argtypes = [C.c_int, C.c_int]
    lib.aki_lab_set_chain_fault.restype = None
    lib.aki_lab_set_chain_fault.argtypes = [C.c_int, C.c_int]
    return lib


class use_lab:
    """Context manager for tests / tools: route every call through the lab library with GEMM tile mode `mode` forced."""

    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        global _lib
        self.saved = _lib
        _lib = load_lab()
        _lib.aki_lab_set_gemm_tile(self.mode)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib.aki_lab_set_gemm_tile(0)
        _lib = self.saved
        return False


class use_lab_attn:
    """Context manager for tests / tools: the lab library with an attention-core variant forced (mma_attn_bf16.hip, g_attn_variant:
    1 = the 32-row kernel at every length, 9 = the 64-row kernel at every length, 10 = the 64-row kernel with the exact running
    maximum, 164 = the 64-row kernel with every tile sent through its exact redo path)."""

    def __init__(self, variant: int):
        self.variant = variant

    def __enter__(self):
        global _lib
        self.saved = _lib
        _lib = load_lab()
        _lib.aki_lab_set_attn_variant(self.variant)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib.aki_lab_set_attn_variant(0)
        _lib = self.saved
        return False


def load() -> C.CDLL:
    """Load the HIP library; raises AkiError (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch ships its own libamdhip64.so.7 and the same SONAME is what this
    # library needs.  Import torch FIRST so both resolve to torch's copy; loading ours first would map
    # /opt/rocm's runtime and leave torch's bundled ROCm stack half-mismatched (launches then fail).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AkiError(f"{LIB_PATH} not found: build it with `python -m aki_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback for the AKI MMA path.")
    _lib = _bind(LIB_PATH)
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().aki_strerror(rc).decode()
        raise AkiError(f"{what or 'aki call'} failed: {msg} (status {rc})")

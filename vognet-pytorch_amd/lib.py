"""ctypes binding of libvog_hip.so (C ABI: include/vog_hip.h).

The product path has NO fallback: if the library is missing or an entry point
fails, an exception is raised. Tensors cross this boundary as raw device
pointers (`tensor.data_ptr()`); torch is only the container.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# VOG_HIP_LIB: an experiment hook (scratch/build_variant.sh builds variants of the library with other compile-time knobs)
LIB_PATH = os.environ.get("VOG_HIP_LIB") or os.path.join(_HERE, "csrc", "libvog_hip.so")

VOG_BF16, VOG_F16 = 0, 1
MDL_KIND = {"igrnd": 0, "vgrnd": 1, "vog": 2}
CONC_TYPE = {"sep": 0, "svsq": 0, "temp": 1, "spat": 2}
DTYPE = {"bf16": VOG_BF16, "f16": VOG_F16, "fp16": VOG_F16}

c_i32, c_i64, c_f32, c_vp = C.c_int, C.c_int64, C.c_float, C.c_void_p


class VogError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("a", c_vp), ("a_is_f32", c_i32), ("lda", c_i64), ("a_rows", c_vp),
                ("w", c_vp), ("ldw", c_i64), ("bias", c_vp), ("residual", c_vp), ("ldr", c_i64),
                ("c32", c_vp), ("c16", c_vp), ("ldc", c_i64), ("ldc16", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("relu", c_i32), ("rep", c_i32),
                ("dtype", c_i32), ("c16_dtype", c_i32), ("out_rows", c_vp), ("out_rows_ncol", c_i32), ("res_vislang", c_vp), ("splitk", c_i32), ("w_frag", c_i32), ("a_frag", c_i32), ("w_lo", c_vp)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.c16_dtype = -1


class SplitkProb(C.Structure):
    _fields_ = [("slabs", c_vp), ("splits", c_i32), ("M", c_i32), ("N", c_i32), ("bias", c_vp),
                ("relu", c_i32), ("rep", c_i32), ("c32", c_vp), ("c16", c_vp), ("ldc", c_i64),
                ("ldc16", c_i64), ("c16_dtype", c_i32)]


class VisprepArgs(C.Structure):
    _fields_ = [("src0", c_vp), ("dst0", c_vp), ("n0", c_i64), ("src1", c_vp), ("dst1", c_vp), ("n1", c_i64),
                ("dtype", c_i32), ("props", c_vp), ("n_rows", c_i32), ("vid_w", c_f32), ("vid_h", c_f32),
                ("w_pe0", c_vp), ("u0", c_vp), ("H0", c_i32), ("nfrm_div0", c_f32),
                ("w_pe1", c_vp), ("u1", c_vp), ("H1", c_i32), ("nfrm_div1", c_f32)]


class QkvArgs(C.Structure):
    _fields_ = [("x16", c_vp), ("ldx", c_i64), ("wqkv", c_vp), ("ldw", c_i64),
                ("q", c_vp), ("k", c_vp), ("vt", c_vp),
                ("S", c_i32), ("N", c_i32), ("H", c_i32), ("dp", c_i32), ("npad", c_i32),
                ("K", c_i32), ("dtype", c_i32),
                ("pl", c_vp), ("nsrl", c_i32), ("nppf", c_i32), ("nfrm", c_i32), ("lang_per_vid", c_i32),
                ("nc_v", c_i32), ("kv_visual_only", c_i32), ("npad_kv", c_i32), ("wqkv_p32", c_vp),
                ("x16_lo", c_vp), ("wqkv_lo", c_vp), ("q_lo", c_vp), ("k_lo", c_vp)]


class AttnStructArgs(C.Structure):
    _fields_ = [("q", c_vp), ("kv", c_vp), ("vv", c_vp), ("pl", c_vp), ("out16", c_vp), ("u", c_vp), ("pe_b", c_vp),
                ("S", c_i32), ("H", c_i32), ("dp", c_i32), ("nsrl", c_i32), ("nppf", c_i32), ("npad_q", c_i32),
                ("npad_kv", c_i32), ("nfrm", c_i32), ("lang_per_vid", c_i32), ("nc_v", c_i32),
                ("use_rel", c_i32), ("seq_per_vid", c_i32), ("NP", c_i32), ("inv_scale", c_f32), ("dtype", c_i32),
                ("q_visual", c_i32), ("guard_flag", c_vp),
                ("q_lo", c_vp), ("kv_lo", c_vp), ("out16_lo", c_vp), ("logit_max", c_vp)]


class QkvCombArgs(C.Structure):
    _fields_ = [("pv", c_vp), ("pl", c_vp), ("q", c_vp), ("k", c_vp), ("vt", c_vp),
                ("n_vid", c_i32), ("nfrm", c_i32), ("nppf", c_i32), ("nsrl", c_i32), ("H", c_i32),
                ("dp", c_i32), ("npad", c_i32), ("lang_per_vid", c_i32), ("nc_v", c_i32),
                ("dtype", c_i32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("vt", c_vp), ("out16", c_vp), ("u", c_vp), ("pe_b", c_vp),
                ("S", c_i32), ("N", c_i32), ("H", c_i32), ("dp", c_i32), ("npad", c_i32),
                ("use_rel", c_i32), ("n_box", c_i32), ("seq_per_vid", c_i32), ("NP", c_i32),
                ("inv_scale", c_f32), ("dtype", c_i32), ("guard_flag", c_vp), ("guard_precleared", c_i32),
                ("q_lo", c_vp), ("k_lo", c_vp), ("out16_lo", c_vp), ("logit_max", c_vp)]


class LstmStepArgs(C.Structure):
    _fields_ = [("gx", c_vp), ("whh", c_vp), ("h_in", c_vp), ("h_out", c_vp), ("c", c_vp),
                ("out16", c_vp), ("lens", c_vp), ("Bn", c_i32), ("T", c_i32), ("R", c_i32),
                ("step", c_i32), ("dtype", c_i32), ("out_frag", c_i32), ("final_row0", c_i32)]


class LstmLayerArgs(C.Structure):
    _fields_ = [("gxs", c_vp), ("whh", c_vp), ("hx", c_vp), ("sync", c_vp), ("out16", c_vp),
                ("lens", c_vp), ("Bn", c_i32), ("T", c_i32), ("R", c_i32), ("dtype", c_i32), ("out_frag", c_i32),
                ("wih", c_vp), ("xa", c_vp), ("bias", c_vp), ("K", c_i32), ("fault", c_vp), ("inject_stall", c_i32),
                ("gx_table", c_vp), ("tok", c_vp)]


class VislangArgs(C.Structure):
    _fields_ = [("vis", c_vp), ("lang", c_vp), ("x32", c_vp), ("x16", c_vp),
                ("n_vid", c_i32), ("nfrm", c_i32), ("nppf", c_i32), ("nsrl", c_i32),
                ("dv", c_i32), ("dl", c_i32), ("lang_per_vid", c_i32), ("nc_v", c_i32),
                ("dtype", c_i32)]


class ScoreArgs(C.Structure):
    _fields_ = [("h1", c_vp), ("w2", c_vp), ("b2", c_vp), ("arg_msk", c_vp), ("cmp_msk", c_vp),
                ("outs", c_vp), ("outs_eval", c_vp),
                ("n_vid", c_i32), ("nfrm", c_i32), ("nppf", c_i32), ("nsrl", c_i32), ("dh", c_i32),
                ("conc_type", c_i32), ("ncmp", c_i32), ("nc_v", c_i32), ("nvl", c_i32),
                ("nfrm0", c_i32), ("nppf0", c_i32)]


class TxTailArgs(C.Structure):
    _fields_ = [("attn16", c_vp), ("kwo", c_i32), ("wo_p", c_vp), ("w1_p", c_vp), ("w2_p", c_vp),
                ("residual", c_vp), ("ldr", c_i64), ("res_vislang", c_vp),
                ("ln1g", c_vp), ("ln1b", c_vp), ("b1", c_vp), ("b2", c_vp), ("ln2g", c_vp), ("ln2b", c_vp),
                ("y32", c_vp), ("y16", c_vp), ("y16_dtype", c_i32), ("wl_p", c_vp), ("bl", c_vp),
                ("score", c_vp), ("head_dtype", c_i32), ("x1_scratch", c_vp),
                ("M", c_i32), ("d", c_i32), ("dh", c_i32), ("dtype", c_i32),
                ("attn16_lo", c_vp), ("wo_p_lo", c_vp), ("w1_p_lo", c_vp), ("w2_p_lo", c_vp), ("y16_lo", c_vp)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.y16_dtype = -1
        self.head_dtype = VOG_F16


class EncoderLayerArgs(C.Structure):
    _fields_ = [("qkv", QkvArgs), ("attn", AttnArgs), ("tail", TxTailArgs)]


class VisencArgs(C.Structure):
    _fields_ = [("prop", c_vp), ("seg", c_vp), ("w_prop_f", c_vp), ("w_seg_f", c_vp), ("b_prop", c_vp),
                ("b_seg", c_vp), ("c32", c_vp), ("c16", c_vp), ("ldc", c_i64), ("c16_dtype", c_i32),
                ("n_prop_rows", c_i32), ("nppf0", c_i32), ("prop_dim", c_i32), ("seg_dim", c_i32),
                ("prop_enc", c_i32), ("seg_enc", c_i32), ("dtype", c_i32), ("lean", c_i32), ("defer_replicas", c_i32),
                ("w_prop_f_lo", c_vp), ("w_seg_f_lo", c_vp), ("c16_lo", c_vp)]


class LossArgs(C.Structure):
    _fields_ = [("mdl_outs", c_vp), ("vidf_outs", c_vp), ("pad_proposals", c_vp), ("pad_gt_bboxs", c_vp),
                ("pad_frm_mask", c_vp), ("pad_pnt_mask", c_vp), ("srl_boxes", c_vp), ("srl_boxes_lens", c_vp),
                ("srl_arg_boxes_mask", c_vp), ("target_cmp", c_vp), ("num_cmp_msk", c_vp), ("verb_cmp", c_vp),
                ("verb_cross_cmp_msk", c_vp), ("out", c_vp), ("scratch", c_vp),
                ("B", c_i32), ("ncmp", c_i32), ("nv", c_i32), ("nsrl", c_i32), ("nbox", c_i32), ("NP", c_i32),
                ("G", c_i32), ("nppf0", c_i32), ("conc_type", c_i32), ("loss_lambda", c_f32)]


class AssembleArgs(C.Structure):
    _fields_ = [("props_in", c_vp), ("props_out", c_vp), ("region_in", c_vp), ("region_out", c_vp),
                ("seg_in", c_vp), ("seg_out", c_vp), ("pnt_in", c_vp), ("pnt_out", c_vp),
                ("gt_in", c_vp), ("gt_out", c_vp), ("num_box", c_vp), ("num_box_out", c_vp), ("target_cmp", c_vp),
                ("srl_boxes_in", c_vp), ("srl_boxes_out", c_vp), ("srl_boxes_lens", c_vp), ("frm_out", c_vp),
                ("B", c_i32), ("ncmp", c_i32), ("nfrm0", c_i32), ("nppf0", c_i32), ("prop_dim", c_i32),
                ("seg_dim", c_i32), ("G", c_i32), ("nv", c_i32), ("nsrl", c_i32), ("nbox", c_i32),
                ("conc_type", c_i32), ("vid_w", c_f32)]


class CopySeg(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("bytes", C.c_size_t)]


MAX_COPY_SEGS = 24
LOGIT_WORDS, LOGIT_STRIDE = 32, 32      # VOG_LOGIT_WORDS / VOG_LOGIT_STRIDE (vog_attn_args.logit_max)


class PredcmpArgs(C.Structure):
    _fields_ = [("final_hidden", c_vp), ("prop_seg", c_vp), ("w0", c_vp), ("b0", c_vp),
                ("w2", c_vp), ("b2", c_vp), ("outs", c_vp), ("arg_msk", c_vp), ("cmp_msk", c_vp),
                ("verb_ind", c_vp), ("vidf_outs", c_vp), ("fin_scores_loss", c_vp),
                ("fin_scores", c_vp),
                ("B", c_i32), ("ncmp", c_i32), ("nvl", c_i32), ("nsrl", c_i32), ("NP", c_i32),
                ("nfrm0", c_i32), ("nppf0", c_i32), ("L", c_i32), ("dp0", c_i32), ("dps", c_i32)]


class PredArgs(C.Structure):
    _fields_ = [("outs_eval", c_vp), ("props", c_vp), ("fin_scores", c_vp), ("rec", c_vp),
                ("B", c_i32), ("ncmp", c_i32), ("nsrl", c_i32), ("nfrm0", c_i32), ("nppf0", c_i32),
                ("conc_type", c_i32), ("logit_max", c_vp), ("stats", c_vp), ("published", c_vp)]


class ModelDesc(C.Structure):
    _fields_ = [("mdl_kind", c_i32), ("conc_type", c_i32),
                ("vocab_size", c_i32), ("emb_dim", c_i32), ("rnn_size", c_i32), ("rnn_layers", c_i32),
                ("prop_dim", c_i32), ("seg_dim", c_i32), ("prop_enc", c_i32), ("seg_enc", c_i32),
                ("lang_enc", c_i32),
                ("obj_layers", c_i32), ("obj_heads", c_i32), ("obj_use_rel", c_i32),
                ("obj_one_frm", c_i32), ("obj_to_use", c_i32),
                ("mul_layers", c_i32), ("mul_heads", c_i32), ("mul_use_rel", c_i32),
                ("nfrm0", c_i32), ("nppf0", c_i32), ("nsrl", c_i32), ("seq_len", c_i32),
                ("vid_w", c_f32), ("vid_h", c_f32), ("tx_dtype", c_i32), ("enc_dtype", c_i32)]


class TailBwdArgs(C.Structure):
    _fields_ = ([(n, c_vp) for n in ("attn", "x", "d_mdl_outs", "d_attn", "d_x",
                                      "wo", "ln1g", "ln1b", "w1", "b1", "w2", "b2", "ln2g", "ln2b", "wl", "bl", "wl2",
                                      "g_wo", "g_ln1g", "g_ln1b", "g_w1", "g_b1", "g_w2", "g_b2", "g_ln2g", "g_ln2b",
                                      "g_wl", "g_bl", "g_wl2", "g_bl2", "scratch")]
                + [("scratch_bytes", C.c_size_t)]
                + [(n, c_i32) for n in ("M", "d", "dh", "dhead", "n_vid", "nfrm", "nppf", "nsrl", "no_head")]
                + [("d_y", c_vp), ("y_out", c_vp), ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", c_i32)])


class LinearF32Args(C.Structure):
    _fields_ = [("x", c_vp), ("ldx", c_i64), ("w", c_vp), ("b", c_vp), ("relu", c_i32), ("y", c_vp),
                ("dy", c_vp), ("ldy", c_i64), ("rep", c_i32), ("g_w", c_vp), ("g_b", c_vp), ("d_x", c_vp),
                ("accumulate_dx", c_i32), ("scratch", c_vp), ("scratch_bytes", C.c_size_t),
                ("M", c_i32), ("N", c_i32), ("K", c_i32)]


_P42 = (c_vp * 2) * 4


class LangF32Args(C.Structure):
    _fields_ = ([(n, c_vp) for n in ("words_ind", "word_mask", "lens", "capture")]
                + [(n, c_i32) for n in ("Bn", "nsrl", "words_len", "mask_len", "T", "vocab_size", "E", "R", "layers", "D", "L")]
                + [("emb", c_vp), ("w_ih", _P42), ("w_hh", _P42), ("b_ih", _P42), ("b_hh", _P42)]
                + [(n, c_vp) for n in ("w_proj", "b_proj", "w_arg", "b_arg", "d_lang_enc", "lang_enc_out", "full_out", "g_emb")]
                + [("g_w_ih", _P42), ("g_w_hh", _P42), ("g_b_ih", _P42), ("g_b_hh", _P42)]
                + [(n, c_vp) for n in ("g_w_proj", "g_b_proj", "g_w_arg", "g_b_arg", "scratch")]
                + [("scratch_bytes", C.c_size_t), ("hid_out", c_vp), ("drop_in", C.c_float), ("drop_out", C.c_float),
                   ("drop_seed", C.c_uint64), ("reuse_forward", c_i32)])


class AttnF32Args(C.Structure):
    _fields_ = [("x", c_vp), ("d_cat", c_vp), ("wq", c_vp), ("wk", c_vp), ("wv", c_vp),
                ("props", c_vp), ("prop_stride", c_i32), ("vid_w", C.c_float), ("vid_h", C.c_float), ("nfrm_div", C.c_float),
                ("pe_w", c_vp), ("pe_b", c_vp), ("cat_out", c_vp),
                ("g_wq", c_vp), ("g_wk", c_vp), ("g_wv", c_vp), ("g_pe_w", c_vp), ("g_pe_b", c_vp),
                ("d_x", c_vp), ("accumulate_dx", c_i32), ("scratch", c_vp), ("scratch_bytes", C.c_size_t),
                ("S", c_i32), ("N", c_i32), ("n", c_i32), ("d", c_i32), ("n_heads", c_i32),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", c_i32)]


class Batch(C.Structure):
    _fields_ = [("B", c_i32), ("ncmp", c_i32), ("T", c_i32),
                ("srl_arg_words_ind", c_vp), ("srl_arg_word_mask", c_vp),
                ("srl_arg_word_mask_len", c_vp), ("srl_arg_words_capture", c_vp),
                ("srl_arg_inds_msk", c_vp), ("num_cmp_msk", c_vp), ("verb_ind_in_srl", c_vp),
                ("pad_region_feature", c_vp), ("seg_feature_for_frms", c_vp), ("pad_proposals", c_vp),
                ("mdl_outs", c_vp), ("mdl_outs_eval", c_vp), ("vidf_outs", c_vp),
                ("fin_scores_loss", c_vp), ("fin_scores", c_vp), ("pred_rec", c_vp),
                ("shared_lang", c_vp), ("shared_final_hidden", c_vp), ("fault", c_vp), ("stats", c_vp)]


# every symbol include/vog_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "vog_version": (c_i32, []),
    "vog_ctx_split_supported": (c_i32, [c_vp, c_i32]),
    "vog_bilstm_fused_cols": (c_i32, []),
    "vog_last_error": (C.c_char_p, []),
    "vog_gemm_bias_act": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "vog_pack_w_frag": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32]),
    "vog_pack_w_frag32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32]),
    "vog_tx_tail_supported": (c_i32, [c_i32, c_i32, c_i32]),
    "vog_tx_tail_scratch_bytes": (c_i64, [c_i32, c_i32]),
    "vog_tx_tail_fwd": (c_i32, [C.POINTER(TxTailArgs), c_vp]),
    "vog_encoder_layer_fwd": (c_i32, [C.POINTER(EncoderLayerArgs), c_vp]),
    "vog_vis_encode_supported": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "vog_vis_encode": (c_i32, [C.POINTER(VisencArgs), c_vp]),
    "vog_seg_replicate": (c_i32, [C.POINTER(VisencArgs), c_vp]),
    "vog_loss_scratch_bytes": (c_i64, [C.POINTER(LossArgs)]),
    "vog_loss_fwd": (c_i32, [C.POINTER(LossArgs), c_vp]),
    "vog_loss_bwd": (c_i32, [C.POINTER(LossArgs), c_vp, c_vp, c_vp]),
    "vog_assemble_batch": (c_i32, [C.POINTER(AssembleArgs), c_vp]),
    "vog_splitk_finish": (c_i32, [C.POINTER(SplitkProb), C.POINTER(SplitkProb), c_vp]),
    "vog_qkv_proj": (c_i32, [C.POINTER(QkvArgs), c_vp]),
    "vog_qkv_rowblock_supported": (c_i32, [c_i32, c_i32]),
    "vog_qkv_combine": (c_i32, [C.POINTER(QkvCombArgs), c_vp]),
    "vog_rel_attention_fwd": (c_i32, [C.POINTER(AttnArgs), c_vp]),
    "vog_rel_attention_struct_fwd": (c_i32, [C.POINTER(AttnStructArgs), c_vp]),
    "vog_residual_layernorm": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vog_cast_f32_to_t16": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "vog_box_u": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp]),
    "vog_srl_gather": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vog_bilstm_layer_supported": (c_i32, [c_i32, c_i32]),
    "vog_mul_tail_bwd_scratch_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "vog_mul_tail_bwd": (c_i32, [C.POINTER(TailBwdArgs), c_vp]),
    "vog_attn_f32_scratch_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "vog_attn_f32": (c_i32, [C.POINTER(AttnF32Args), c_vp]),
    "vog_conc_f32_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 8 + [c_vp]),
    "vog_linear_f32_scratch_bytes": (c_i64, [c_i32, c_i32]),
    "vog_linear_f32": (c_i32, [C.POINTER(LinearF32Args), c_vp]),
    "vog_lang_f32_scratch_bytes": (c_i64, [c_i32] * 8),
    "vog_lang_f32": (c_i32, [C.POINTER(LangF32Args), c_vp]),
    "vog_concat_rows_f32": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "vog_conc_f32_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 8 + [c_vp]),
    "vog_score_head_f32": (c_i32, [c_vp] * 7 + [C.c_size_t] + [c_i32] * 7 + [c_vp]),
    "vog_adam_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, c_i32, c_vp]),
    "vog_row_mean_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vog_train_set_int": (c_i32, [C.c_char_p, c_i32]),
    "vog_train_get_int": (c_i32, [C.c_char_p, C.POINTER(c_i32)]),
    "vog_bilstm_fwd": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, c_vp, c_vp, c_vp]),
    "vog_lstm_out_to_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vog_score_head_f32_bwd_scratch_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "vog_score_head_f32_bwd": (c_i32, [c_vp] * 11 + [C.c_size_t] + [c_i32] * 7 + [c_vp]),
    "vog_bilstm_hx_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "vog_bilstm_layer": (c_i32, [C.POINTER(LstmLayerArgs), c_vp]),
    "vog_lstm_schedule": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "vog_lstm_pack_whh": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32]),
    "vog_lstm_pack_w": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32]),
    "vog_prep_fused": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                               c_vp, c_vp, c_i32, C.POINTER(VisprepArgs), c_vp]),
    "vog_lang_prep": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                              c_vp, c_vp, c_i32, c_vp]),
    "vog_vis_prep": (c_i32, [C.POINTER(VisprepArgs), c_vp]),
    "vog_bilstm_step": (c_i32, [C.POINTER(LstmStepArgs), c_vp]),
    "vog_srl_argvec": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vog_vislang_layout": (c_i32, [C.POINTER(VislangArgs), c_vp]),
    "vog_score_head": (c_i32, [C.POINTER(ScoreArgs), c_vp]),
    "vog_pred_cmp_head": (c_i32, [C.POINTER(PredcmpArgs), c_vp]),
    "vog_pred_record_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "vog_pred_head": (c_i32, [C.POINTER(PredArgs), c_vp]),
    "vog_ctx_create": (c_i32, [C.POINTER(ModelDesc), C.POINTER(c_vp)]),
    "vog_ctx_set_weight": (c_i32, [c_vp, C.c_char_p, c_vp, c_i64]),
    "vog_ctx_finalize": (c_i32, [c_vp]),
    "vog_ctx_destroy": (c_i32, [c_vp]),
    "vog_ctx_num_weights": (c_i32, [c_vp]),
    "vog_ctx_weight_name": (C.c_char_p, [c_vp, c_i32]),
    "vog_ctx_weight_numel": (c_i64, [c_vp, c_i32]),
    "vog_workspace_bytes": (c_i64, [c_vp, c_i32, c_i32, c_i32]),
    "vog_workspace_init": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, C.c_size_t, c_vp]),
    "vog_forward": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, c_vp]),
    "vog_workspace_stage": (c_i32, [c_vp, c_i32, c_i32, c_i32, C.c_char_p, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "vog_graph_capture": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, c_vp, C.POINTER(c_vp)]),
    "vog_graph_capture_fed": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, C.POINTER(CopySeg), C.POINTER(AssembleArgs),
                                      C.POINTER(CopySeg), c_i32, c_vp, C.POINTER(c_vp)]),
    "vog_copy_segments": (c_i32, [C.POINTER(CopySeg), c_i32, c_vp]),
    "vog_ctx_set_int": (c_i32, [c_vp, C.c_char_p, c_i32]),
    "vog_graph_launch": (c_i32, [c_vp, c_vp]),
    "vog_graph_destroy": (c_i32, [c_vp]),
    "vog_lang_workspace_bytes": (c_i64, [c_vp, c_i32, c_i32, c_i32]),
    "vog_lang_workspace_init": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, C.c_size_t, c_vp]),
    "vog_lang_forward": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, c_vp]),
    "vog_lang_outputs": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, C.POINTER(c_vp), C.POINTER(c_vp)]),
    "vog_group_forward": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, C.POINTER(C.POINTER(Batch)),
                                  C.POINTER(c_vp), C.POINTER(C.c_size_t), c_i32, c_vp]),
    "vog_group_graph_capture": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, C.POINTER(C.POINTER(Batch)),
                                        C.POINTER(c_vp), C.POINTER(C.c_size_t), c_i32, c_vp, C.POINTER(c_vp)]),
    "vog_time_kernel": (c_i32, [c_vp, C.POINTER(Batch), c_vp, C.c_size_t, C.c_char_p, c_i32, c_vp, C.POINTER(c_f32)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libvog_hip.so and type every entry point. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; load it FIRST so that libvog_hip binds to
    # the same HIP runtime (a second runtime instance sees no devices / no streams)
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise VogError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
            "(or vognet-pytorch_amd/csrc/build.py). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vog_last_error()
        raise VogError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> Optional[int]:
    """torch tensor (or None) -> raw pointer value."""
    if t is None:
        return None
    assert t.is_contiguous(), "libvog_hip takes contiguous tensors"
    return t.data_ptr()


def stream_ptr(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream

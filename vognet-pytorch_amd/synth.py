"""Synthetic batches and seeded weights for the VOGNet forward path.

There is no dataset here (reference data is ~530 GB, data/README.md:19,43), so
parity tests, goldens and bench.py all draw from this generator. It is
platform independent (numpy PCG64 via `default_rng`), so the same seed gives
the same tensors in this container and on the GPU box.

Batch schema = what the reference `BatchCollator` hands to `model(batch)`
(code/dat_loader_simple.py:1530-1544; key list: SURVEY.md section 8(b)):

  spat / temp (one model "video" per query, language axis nv = 1)
    srl_arg_words_ind   [B,1,5,20] i64   word ids per argument slot
    srl_arg_word_mask   [B,1,20]   i64   arg*20+w of the t-th sentence word, -1 pad
    srl_tag_word_ind    [B,1,20]   i64   (shape-checked only)
    srl_arg_word_mask_len [B,1]    i64   sentence length
    srl_arg_words_capture [B,1,5,2] i64  first/last sentence position of each arg
    srl_arg_inds_msk    [B,1,5]    i64   1 = real argument
    pad_region_feature  [B,NP,2048] f32  NP = ncmp*10*nppf0
    seg_feature_for_frms [B,ncmp*10,3072] f32
    pad_proposals       [B,NP,7]   f32   x1,y1,x2,y2,frame,class,score
    new_srl_idxs        [B,ncmp]   i64   (size only)
    num_cmp_msk         [B,ncmp]   i64
  sep / svsq: language keys carry a leading [B,ncmp,...] (the sentence copied
    into every video), visual keys are [B,ncmp,10*nppf0,...], plus
    verb_ind_in_srl [B,ncmp].

Row order of the visual keys follows the loader: temp = (video, frame, prop)
with frame index += 10*video (dat_loader_simple.py:1231-1252); spat =
(frame, video, prop) with x += 720*video (:1067-1103); sep = per video.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

VID_W = 720.0
VID_H = 405.0
NSRL = 5
SEQ_LEN = 20
NFRM0 = 10

DEFAULT_ARG_LENS = (3, 1, 4, 4, 0)   # src_len 12: spans [0,2],[3,3],[4,7],[8,11] + masked slot


def _lang_one(rng, vocab_size: int, arg_lens: Sequence[int]):
    """Language tensors of ONE query (loader: dat_loader_simple.py:623-656)."""
    words = np.zeros((NSRL, SEQ_LEN), np.int64)
    mask = np.full((SEQ_LEN,), -1, np.int64)
    capture = np.zeros((NSRL, 2), np.int64)
    inds_msk = np.zeros((NSRL,), np.int64)
    pos = 0
    for a, L in enumerate(arg_lens):
        if L <= 0:
            continue
        words[a, :L] = rng.integers(0, vocab_size, size=L)
        mask[pos:pos + L] = a * SEQ_LEN + np.arange(L)
        capture[a] = (pos, pos + L - 1)
        inds_msk[a] = 1
        pos += L
    assert pos <= SEQ_LEN
    return words, mask, pos, capture, inds_msk


def ragged_arg_lens(rng, lo: int = 6, hi: int = 18) -> List[int]:
    """Random argument lengths with total sentence length in [lo, hi]."""
    total = int(rng.integers(lo, hi + 1))
    nargs = int(rng.integers(2, NSRL + 1))
    cuts = np.sort(rng.choice(np.arange(1, total), size=nargs - 1, replace=False))
    lens = np.diff(np.concatenate([[0], cuts, [total]])).tolist()
    lens += [0] * (NSRL - nargs)
    return [int(x) for x in lens]


def make_batch(conc_type: str, B: int, nppf0: int, *, ncmp: int = 4,
               vocab_size: int = 5000, prop_dim: int = 2048,
               seg_dim: int = 3072, seed: int = 0, ragged: bool = False,
               num_cmp_msk: Optional[np.ndarray] = None,
               arg_lens: Optional[Sequence[Sequence[int]]] = None
               ) -> Dict[str, np.ndarray]:
    """One synthetic batch (numpy, fp32/int64) of SURVEY.md section 8(d)."""
    assert conc_type in ("spat", "temp", "sep", "svsq")
    if conc_type == "svsq":
        ncmp = 1
    rng = np.random.default_rng(seed)
    sep = conc_type in ("sep", "svsq")
    nv = ncmp if sep else 1

    W = np.zeros((B, nv, NSRL, SEQ_LEN), np.int64)
    M = np.zeros((B, nv, SEQ_LEN), np.int64)
    L = np.zeros((B, nv), np.int64)
    C = np.zeros((B, nv, NSRL, 2), np.int64)
    I = np.zeros((B, nv, NSRL), np.int64)
    V = np.zeros((B, ncmp), np.int64)
    for b in range(B):
        if arg_lens is not None:          # explicit per-query argument lengths (edge cases)
            lens = list(arg_lens[b % len(arg_lens)])
        else:
            lens = ragged_arg_lens(rng) if ragged else list(DEFAULT_ARG_LENS)
        w, m, l, c, i = _lang_one(rng, vocab_size, lens)
        W[b, :], M[b, :], L[b, :], C[b, :], I[b, :] = w, m, l, c, i
        nreal = int(i.sum())
        V[b, :] = int(rng.integers(0, nreal))
    tags = rng.integers(0, 10, size=(B, nv, SEQ_LEN)).astype(np.int64)

    npv = NFRM0 * nppf0                     # proposals of one source video
    vis_lead = (B, ncmp) if sep else (B,)
    NP = npv if sep else ncmp * npv
    F = NFRM0 if sep else ncmp * NFRM0

    feats = rng.standard_normal(vis_lead + (NP, prop_dim), dtype=np.float32)
    segs = rng.standard_normal(vis_lead + (F, seg_dim), dtype=np.float32)

    # boxes of every source video: [B, ncmp, frame, prop, 7]
    x = np.sort(rng.uniform(0, VID_W, size=(B, ncmp, NFRM0, nppf0, 2)), axis=-1)
    y = np.sort(rng.uniform(0, VID_H, size=(B, ncmp, NFRM0, nppf0, 2)), axis=-1)
    box = np.zeros((B, ncmp, NFRM0, nppf0, 7), np.float32)
    box[..., 0], box[..., 2] = x[..., 0], x[..., 1]
    box[..., 1], box[..., 3] = y[..., 0], y[..., 1]
    box[..., 4] = np.arange(NFRM0, dtype=np.float32)[None, None, :, None]
    box[..., 5] = rng.integers(0, 431, size=box.shape[:-1]).astype(np.float32)
    box[..., 6] = rng.uniform(0, 1, size=box.shape[:-1]).astype(np.float32)
    if conc_type == "temp":
        box[..., 4] += (NFRM0 * np.arange(ncmp, dtype=np.float32))[None, :, None, None]
        props = box.reshape(B, NP, 7)                       # (video, frame, prop)
    elif conc_type == "spat":
        off = (VID_W * np.arange(ncmp, dtype=np.float32))[None, :, None, None]
        box[..., 0] += off
        box[..., 2] += off
        props = box.transpose(0, 2, 1, 3, 4).reshape(B, NP, 7)   # (frame, video, prop)
    else:
        props = box.reshape(B, ncmp, NP, 7)

    if num_cmp_msk is None:
        num_cmp_msk = np.ones((B, ncmp), np.int64)
    out = {
        "srl_arg_words_ind": W,
        "srl_arg_word_mask": M,
        "srl_tag_word_ind": tags,
        "srl_arg_word_mask_len": L,
        "srl_arg_words_capture": C,
        "srl_arg_inds_msk": I,
        "pad_region_feature": feats,
        "seg_feature_for_frms": segs,
        "pad_proposals": np.ascontiguousarray(props, dtype=np.float32),
        "new_srl_idxs": rng.integers(0, 1000, size=(B, ncmp)).astype(np.int64),
        "num_cmp_msk": np.asarray(num_cmp_msk, np.int64).reshape(B, ncmp),
    }
    if sep:
        out["verb_ind_in_srl"] = V
    return out


# --------------------------------------------------------------------------- #
# weights
# --------------------------------------------------------------------------- #
def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _linear(sd, rng, name, out_f, in_f, bias=True):
    k = 1.0 / math.sqrt(in_f)
    sd[name + ".weight"] = _uniform(rng, (out_f, in_f), k)
    if bias:
        sd[name + ".bias"] = _uniform(rng, (out_f,), k)


def _tx(sd, rng, prefix, d, n_layers, perturb_ln):
    dh = d // 2
    for l in range(n_layers):
        p = f"{prefix}.encoder.layers.{l}"
        for w in ("wq", "wk", "wv", "wo"):
            _linear(sd, rng, f"{p}.selfattn.layer.{w}", d, d, bias=False)
        _linear(sd, rng, f"{p}.feedforward.layer.linear1", dh, d)
        _linear(sd, rng, f"{p}.feedforward.layer.linear2", d, dh)
        for blk in ("selfattn", "feedforward"):
            g = np.ones((d,), np.float32)
            b = np.zeros((d,), np.float32)
            if perturb_ln:
                g += 0.1 * rng.standard_normal(d).astype(np.float32)
                b += 0.1 * rng.standard_normal(d).astype(np.float32)
            sd[f"{p}.{blk}.layernorm.weight"] = g
            sd[f"{p}.{blk}.layernorm.bias"] = b


def init_state_dict(cfg, vocab_size: int, seed: int = 0,
                    perturb_ln: bool = False) -> Dict[str, np.ndarray]:
    """Seeded, torch-default-init-like weights under the reference's
    state-dict key names (SURVEY.md section 8(b) "Checkpoint"; reference module
    construction: code/mdl_vog.py:156-230, 412-451, 548-585).
    """
    rng = np.random.default_rng(seed)
    m = cfg.mdl
    E, R, NL = m.input_encoding_size, m.rnn.rnn_size, m.rnn.num_layers
    pe, se, le = (m.vsrl.prop_encode_size, m.vsrl.seg_encode_size,
                  m.vsrl.lang_encode_size)
    sd: Dict[str, np.ndarray] = {}
    emb = rng.standard_normal((vocab_size + 1, E)).astype(np.float32)
    emb[vocab_size] = 0.0                               # padding_idx row
    sd["lstm_encoder.embed_tokens.weight"] = emb
    k = 1.0 / math.sqrt(R)
    for l in range(NL):
        in_f = E if l == 0 else 2 * R
        for sfx in ("", "_reverse"):
            sd[f"lstm_encoder.lstm.weight_ih_l{l}{sfx}"] = _uniform(rng, (4 * R, in_f), k)
            sd[f"lstm_encoder.lstm.weight_hh_l{l}{sfx}"] = _uniform(rng, (4 * R, R), k)
            sd[f"lstm_encoder.lstm.bias_ih_l{l}{sfx}"] = _uniform(rng, (4 * R,), k)
            sd[f"lstm_encoder.lstm.bias_hh_l{l}{sfx}"] = _uniform(rng, (4 * R,), k)
    _linear(sd, rng, "lstm_out_feat_proj.0", le, 2 * R)
    _linear(sd, rng, "srl_arg_words_out_enc.0", le, 2 * le)
    _linear(sd, rng, "srl_simple_lin.0", le, 3 * le)
    _linear(sd, rng, "prop_encoder.0", pe, m.prop_feat_dim)
    _linear(sd, rng, "seg_encoder.0", se, m.seg_feat_dim)
    _linear(sd, rng, "seg_verb_classf.0", 256, se + le)
    _linear(sd, rng, "seg_verb_classf.2", 1, 256)
    d_obj = pe + se
    d_mul = d_obj + le
    _linear(sd, rng, "lin2.0", 256, d_mul)
    _linear(sd, rng, "lin2.2", 1, 256)
    _linear(sd, rng, "lin_tmp.0", 256, d_mul)
    _linear(sd, rng, "lin_tmp.2", 1, 256)
    if m.name in ("vgrnd", "vog"):
        _tx(sd, rng, "obj_txf", d_obj, m.obj_tx.n_layers, perturb_ln)
        _linear(sd, rng, "pe_obj_sub_enc.0", m.obj_tx.n_heads, 5)
    if m.name == "vog":
        _tx(sd, rng, "mult_txf", d_mul, m.mul_tx.n_layers, perturb_ln)
        _linear(sd, rng, "pe_mul_sub_enc.0", m.mul_tx.n_heads, 5)
    return sd


# --------------------------------------------------------------------------- #
# loss-side keys of a loader batch (dat_loader_simple.py:425-459,
# 734-780, 1476-1505): ground-truth boxes, frame / padding masks, per-argument box
# indices and the contrastive-sampling targets. Own RNG stream: make_batch's draws (and
# the SHA-256 of its inputs stored in the forward goldens) are unchanged.
# --------------------------------------------------------------------------- #
def make_targets(batch: Dict[str, np.ndarray], conc_type: str, nppf0: int, *, n_gt: int = 100,
                 n_box: int = 4, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(10_007 + seed)
    sep = conc_type in ("sep", "svsq")
    props = batch["pad_proposals"]
    B, ncmp = batch["num_cmp_msk"].shape
    lead = props.shape[:-2]
    NP = props.shape[-2]
    nv = batch["srl_arg_inds_msk"].shape[1]
    nsrl = batch["srl_arg_inds_msk"].shape[2]
    flat = props.reshape(-1, NP, 7)
    n = flat.shape[0]
    gt = np.zeros((n, n_gt, 5), np.float32)
    frm = np.zeros((n, NP, n_gt), np.uint8)
    pnt = np.zeros((n, NP), np.uint8)
    for i in range(n):
        # a third of the gt slots are jittered copies of proposals (IoU well above 0.5 with their
        # source), a third random boxes, the rest empty padding (x1=y1=x2=y2=0 -> "area zero")
        src = rng.integers(0, NP, size=n_gt)
        k1, k2 = n_gt // 3, 2 * n_gt // 3
        g = flat[i, src, :5].copy()
        g[:k1, :4] += rng.uniform(-3, 3, size=(k1, 4)).astype(np.float32)
        x = np.sort(rng.uniform(0, VID_W, size=(k2 - k1, 2)), axis=-1)
        y = np.sort(rng.uniform(0, VID_H, size=(k2 - k1, 2)), axis=-1)
        g[k1:k2, 0], g[k1:k2, 2], g[k1:k2, 1], g[k1:k2, 3] = x[:, 0], x[:, 1], y[:, 0], y[:, 1]
        g[k2:] = 0.0
        gt[i] = g
        # the mask multiplies the IoU (utils/box_utils.py:108-109): 1 where proposal and gt box share a
        # frame, plus a few random entries; pnt marks a handful of padded proposals
        frm[i] = (flat[i, :, 4][:, None] == g[None, :, 4]).astype(np.uint8)
        frm[i] |= (rng.uniform(size=(NP, n_gt)) < 0.02).astype(np.uint8)
        pnt[i] = (rng.uniform(size=NP) < 0.03).astype(np.uint8)
        # a few degenerate proposals are needed to hit the "anchor area zero -> -1" branch, but the
        # proposals are inputs of the forward and stay untouched: degenerate GT boxes cover the other branch
    srl_boxes = rng.integers(0, n_gt // 3, size=(B, nv, nsrl, n_box)).astype(np.int64)
    srl_boxes_lens = (rng.uniform(size=(B, nv, nsrl, n_box)) < 0.7).astype(np.int64)
    arg_boxes_mask = (batch["srl_arg_inds_msk"] * (rng.uniform(size=(B, nv, nsrl)) < 0.8)).astype(np.int64)
    out = {
        "pad_gt_bboxs": gt.reshape(lead + (n_gt, 5)),
        "pad_frm_mask": frm.reshape(lead + (NP, n_gt)),
        "pad_pnt_mask": pnt.reshape(lead + (NP,)),
        "srl_boxes": srl_boxes, "srl_boxes_lens": srl_boxes_lens, "srl_arg_boxes_mask": arg_boxes_mask,
        "target_cmp": rng.integers(0, ncmp, size=(B,)).astype(np.int64),
    }
    if sep:
        out["verb_cmp"] = (rng.uniform(size=(B, ncmp)) < 0.5).astype(np.int64)
        out["verb_cross_cmp_msk"] = (rng.uniform(size=(B, ncmp, ncmp)) < 0.6).astype(np.int64)
    return out


# --------------------------------------------------------------------------- #
# per-video ITEMS of a query as `AV_CS.itemcollector` stacks them (dat_loader_simple.py:1405-1510,
# SURVEY.md App. B.5): the input of the SPAT / TEMP batch assembly (verb_item_getter_SPAT / _TEMP,
# dat_loader_simple.py:1046-1338). Leading axes [B, ncmp].
# --------------------------------------------------------------------------- #
def make_items(B: int, ncmp: int, nppf0: int, *, prop_dim: int = 2048, seg_dim: int = 3072, n_gt: int = 100,
               nsrl: int = NSRL, n_box: int = 4, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(20_011 + seed)
    NPv = NFRM0 * nppf0
    x = np.sort(rng.uniform(0, VID_W, size=(B, ncmp, NFRM0, nppf0, 2)), axis=-1)
    y = np.sort(rng.uniform(0, VID_H, size=(B, ncmp, NFRM0, nppf0, 2)), axis=-1)
    box = np.zeros((B, ncmp, NFRM0, nppf0, 7), np.float32)
    box[..., 0], box[..., 2], box[..., 1], box[..., 3] = x[..., 0], x[..., 1], y[..., 0], y[..., 1]
    box[..., 4] = np.arange(NFRM0, dtype=np.float32)[None, None, :, None]
    box[..., 5] = rng.integers(0, 431, size=box.shape[:-1]).astype(np.float32)
    box[..., 6] = rng.uniform(0, 1, size=box.shape[:-1]).astype(np.float32)
    num_box = rng.integers(0, 12, size=(B, ncmp)).astype(np.int64)
    num_box[0, 0] = 0                                      # a video without ground-truth boxes
    if B > 1:
        num_box[1, :] = 0                                  # a query without any (the "gt_boxs[0, 0]" fallback)
    gt = np.zeros((B, ncmp, n_gt, 5), np.float32)
    gx = np.sort(rng.uniform(0, VID_W, size=(B, ncmp, n_gt, 2)), axis=-1)
    gy = np.sort(rng.uniform(0, VID_H, size=(B, ncmp, n_gt, 2)), axis=-1)
    gt[..., 0], gt[..., 2], gt[..., 1], gt[..., 3] = gx[..., 0], gx[..., 1], gy[..., 0], gy[..., 1]
    gt[..., 4] = rng.integers(0, NFRM0, size=(B, ncmp, n_gt)).astype(np.float32)
    return {
        "pad_proposals": box.reshape(B, ncmp, NPv, 7),
        "pad_region_feature": rng.standard_normal((B, ncmp, NPv, prop_dim), dtype=np.float32),
        "seg_feature_for_frms": rng.standard_normal((B, ncmp, NFRM0, seg_dim), dtype=np.float32),
        "pad_pnt_mask": (rng.uniform(size=(B, ncmp, NPv)) < 0.1).astype(np.uint8),
        "pad_gt_bboxs": gt, "num_box": num_box,
        "target_cmp": rng.integers(0, ncmp, size=(B,)).astype(np.int64),
        "srl_boxes": rng.integers(0, 12, size=(B, 1, nsrl, n_box)).astype(np.int64),
        "srl_boxes_lens": (rng.uniform(size=(B, 1, nsrl, n_box)) < 0.6).astype(np.int64),
    }


def sharpen_state_dict(sd: Dict[str, np.ndarray], qk: float, pe: float = 4.0) -> Dict[str, np.ndarray]:
    """Weights away from the random-init regime (in place): wq / wk of every obj_txf / mult_txf layer x qk (attention logits
    x qk^2: x 8 -> std ~1 nat, x 16 -> ~4-30), the box-bias Linear(5, H) x pe. What a trained checkpoint (EXPTS.md:95-189) looks like
    to the precision plan (engine.attention_sharpness); used by the sharpened parity cases (oracle/cases.py) and bench.py's
    `hi_lo_plan` measurement."""
    for k in list(sd):
        if k.endswith("selfattn.layer.wq.weight") or k.endswith("selfattn.layer.wk.weight"):
            sd[k] = (sd[k] * np.float32(qk)).astype(np.float32)
        elif k.startswith("pe_obj_sub_enc.") or k.startswith("pe_mul_sub_enc."):
            sd[k] = (sd[k] * np.float32(pe)).astype(np.float32)
    return sd

"""CPU oracle for the VOGNet forward path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch restatement (torch-CPU fp32 tensor ops, no nn.Module, no
nn.LSTM) of the reference algorithm on the hot path, written from the
semantics in SURVEY.md section 8(a) / Appendix B. Each function cites the reference
file:line it restates. Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product path
(`vognet-pytorch_amd/`) never does.

Parity status: PINNED. The reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against the reference itself imported in
the build container (`oracle/ref_import.py`, `oracle/make_golden.py`) — outputs
committed as fixtures under `tests/golden/` and re-checked on every CPU test
run (`tests/test_oracle_golden.py`); when `/root/reference` is present
`tests/test_oracle_vs_reference.py` additionally compares live.

`quant` hooks: `forward(..., quant=fn)` rounds the operands of the matrix
contractions in the named scopes through `fn` (e.g. fp32->bf16->fp32). They
exist only to size the tolerance budget of the 16-bit MFMA path in tests.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch

F32 = torch.float32


# --------------------------------------------------------------------------- #
# configuration of one model instance (what the reference reads from cfg/comm)
# --------------------------------------------------------------------------- #
@dataclass
class OracleCfg:
    mdl_name: str = "vog"            # igrnd | vgrnd | vog     (mdl_selector.py:26-69)
    conc_type: str = "spat"          # sep | svsq | temp | spat
    vocab_size: int = 5000
    nppf0: int = 5                   # comm.num_prop_per_frm
    nfrm0: int = 10                  # cfg.ds.num_sampled_frm
    vid_w: float = 720.0
    vid_h: float = 405.0
    rnn_layers: int = 2
    obj_layers: int = 1
    obj_heads: int = 3
    obj_use_rel: bool = True
    obj_one_frm: bool = False
    obj_to_use: bool = True
    mul_layers: int = 1
    mul_heads: int = 3
    mul_use_rel: bool = True

    @staticmethod
    def from_cfg(cfg, vocab_size: int, nppf0: int) -> "OracleCfg":
        m = cfg.mdl
        return OracleCfg(
            mdl_name=m.name, conc_type=cfg.ds.conc_type, vocab_size=vocab_size,
            nppf0=nppf0, nfrm0=cfg.ds.num_sampled_frm,
            vid_w=float(cfg.ds.resized_width), vid_h=float(cfg.ds.resized_height),
            rnn_layers=m.rnn.num_layers,
            obj_layers=m.obj_tx.n_layers, obj_heads=m.obj_tx.n_heads,
            obj_use_rel=m.obj_tx.use_rel, obj_one_frm=m.obj_tx.one_frm,
            obj_to_use=m.obj_tx.to_use,
            mul_layers=m.mul_tx.n_layers, mul_heads=m.mul_tx.n_heads,
            mul_use_rel=m.mul_tx.use_rel)


def _q(quant, scope, *xs):
    """Apply the test-only operand rounding hook."""
    if quant is None:
        return xs if len(xs) > 1 else xs[0]
    ys = tuple(quant(scope, x) for x in xs)
    return ys if len(ys) > 1 else ys[0]


# --------------------------------------------------------------------------- #
# language side
# --------------------------------------------------------------------------- #
def drop_mask(seed: int, site: int, shape, p: float) -> torch.Tensor:
    """Dropout multipliers (0 or 1 / (1 - p)) of the counter-based generator the DEVICE training path uses
    (csrc/backward.hip::drop_scale): element idx (row-major) of site `site` is kept iff the top 32 bits of
    splitmix64(seed * 0x9E3779B97F4A7C15 + site * 0xBF58476D1CE4E5B9 + idx) are >= p * 2^32. The reference draws its masks
    from torch's generator (nn.Dropout / F.dropout, transformer_code.py:26-50, utils/mdl_srl_utils.py:128-150), which no
    other implementation can reproduce; with THESE masks the oracle's autograd is what the device must equal."""
    import numpy as np
    n = int(np.prod(shape))
    p32 = float(np.float32(p))
    thr = np.uint64(int(p32 * 4294967296.0))
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(site) * np.uint64(0xBF58476D1CE4E5B9)
             + np.arange(n, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    keep = (z >> np.uint64(32)) >= thr
    inv = np.float32(1.0 / (1.0 - p32))
    return torch.from_numpy(np.where(keep, inv, np.float32(0)).astype(np.float32).reshape(shape))


# dropout sites (the same numbers in vognet-pytorch_amd/train.py): language 1 = embeddings, 2 + l = behind BiLSTM layer l
# (inter-layer dropout of nn.LSTM), 10 = LSTM output; transformer stack k (1 = obj_tx, 2 = mul_tx), layer l: 100 k + 10 l + {0: attention
# probabilities [S, H, N, N], 1: the attention sub-layer's output, 2: the feed-forward sub-layer's output}
def _drop(drop, site, x, p):
    if drop is None or p <= 0.0:
        return x
    return x * drop_mask(drop, site, x.shape, p)


def srl_arg_seq_to_sent_seq(words_ind, word_mask, vocab_size):
    """Token re-index (reference mdl_vog.py:67-95).
    tok[b,t] = words[b, m[b,t]] if m[b,t] >= 0 else V. Does NOT mutate inputs."""
    B, nv, nsrl, sl = words_ind.shape
    words = words_ind.reshape(B * nv, nsrl * sl)
    m = word_mask.reshape(B * nv, -1)
    pad = m < 0
    tok = torch.gather(words, 1, m.clamp(min=0))
    tok = torch.where(pad, torch.full_like(tok, vocab_size), tok)
    return tok


def _lstm_dir(xg, w_hh, lens, reverse, quant):
    """One direction of one layer with packed-sequence semantics
    (reference utils/mdl_srl_utils.py:134-148 via nn.LSTM + pack/pad; gate
    order i,f,g,o). xg = x W_ih^T + b_ih + b_hh, [Bn,T,4R]."""
    Bn, T, G = xg.shape
    R = G // 4
    h = torch.zeros(Bn, R, dtype=F32)
    c = torch.zeros(Bn, R, dtype=F32)
    out = torch.zeros(Bn, T, R, dtype=F32)
    ar = torch.arange(Bn)
    w_hh_t = _q(quant, "lstm", w_hh).t()
    for s in range(T):
        active = s < lens                                   # [Bn]
        pos = (lens - 1 - s).clamp(min=0) if reverse else torch.full_like(lens, s)
        g = xg[ar, pos] + _q(quant, "lstm", h) @ w_hh_t
        i, f, gg, o = g.split(R, dim=1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        a = active.unsqueeze(1)
        c = torch.where(a, c_new, c)
        h = torch.where(a, h_new, h)
        idx = torch.nonzero(active).squeeze(1)
        out[idx, pos[idx]] = h_new[idx]
    return out, h


def lstm_encoder(tokens, lens, sd, num_layers, quant=None, drop=None, p_in=0.1, p_out=0.1):
    """Embedding + packed multi-layer BiLSTM (reference
    utils/mdl_srl_utils.py:114-169).
    Returns (x [Bn,T,2R] zero past each length, final_hidden_last [Bn,2R] =
    [h_fwd(last valid) || h_bwd(step 0)] of the top layer)."""
    emb = sd["lstm_encoder.embed_tokens.weight"]
    x = _drop(drop, 1, emb[tokens], p_in)                    # [Bn,T,E]   (F.dropout(x, dropout_in), mdl_srl_utils.py:128)
    hf = hb = None
    for l in range(num_layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            p = "lstm_encoder.lstm."
            w_ih = sd[f"{p}weight_ih_l{l}{sfx}"]
            w_hh = sd[f"{p}weight_hh_l{l}{sfx}"]
            b = sd[f"{p}bias_ih_l{l}{sfx}"] + sd[f"{p}bias_hh_l{l}{sfx}"]
            xq, wq = _q(quant, "lstm", x, w_ih)
            xg = xq @ wq.t() + b
            o, h = _lstm_dir(xg, w_hh, lens, rev, quant)
            outs.append(o)
            if rev:
                hb = h
            else:
                hf = h
        x = torch.cat(outs, dim=2)
        # nn.LSTM(dropout=dropout_out) between the layers, F.dropout(x, dropout_out) on the output (mdl_srl_utils.py:104,150)
        x = _drop(drop, 2 + l if l < num_layers - 1 else 10, x, p_out)
    return x, torch.cat([hf, hb], dim=1)


def linear(x, sd, name, relu=False, quant=None, scope="enc"):
    w = sd[name + ".weight"]
    xq, wq = _q(quant, scope, x, w)
    y = xq @ wq.t()
    if (name + ".bias") in sd:
        y = y + sd[name + ".bias"]
    return torch.relu(y) if relu else y


def lang_encode(tokens, lens, sd, num_layers, quant=None, drop=None):
    """reference mdl_vog.py:250-283: truncate to T = max len, LSTM, then
    lstm_out_feat_proj on every step and on final_hidden[-1]."""
    T = int(lens.max().item())
    x, fin = lstm_encoder(tokens[:, :T].contiguous(), lens, sd, num_layers, quant, drop=drop)
    full = linear(x, sd, "lstm_out_feat_proj.0", relu=True, quant=quant, scope="enc.lang")
    hid = linear(fin, sd, "lstm_out_feat_proj.0", relu=True, quant=quant, scope="enc.lang")
    return full, hid


def retrieve_srl_args(full, capture, inds_msk, sd, quant=None, stash=None):
    """reference mdl_vog.py:97-140."""
    B, nv, nsrl, _ = capture.shape
    cap = capture.reshape(B * nv, nsrl, 2)
    D = full.shape[-1]
    st = torch.gather(full, 1, cap[..., 0].unsqueeze(-1).expand(-1, -1, D))
    en = torch.gather(full, 1, cap[..., 1].unsqueeze(-1).expand(-1, -1, D))
    enc = torch.cat([st, en], dim=2).reshape(B, nv, nsrl, 2 * D)
    out = linear(enc, sd, "srl_arg_words_out_enc.0", relu=True, quant=quant)
    if stash is not None:
        stash["lang_enc"] = out      # before the argument mask (backward fixtures)
    return out * inds_msk.unsqueeze(-1).to(F32)


# --------------------------------------------------------------------------- #
# transformer
# --------------------------------------------------------------------------- #
def chunk_sizes(d, n):
    """torch.chunk split sizes (transformer_code.py:66-67,182-183): ceil(d/n)
    each, last one shorter -> 512/3 = 171,171,170."""
    c = -(-d // n)
    out = []
    r = d
    while r > 0:
        out.append(min(c, r))
        r -= c
    assert len(out) == n
    return out


def normalise_boxes(props5, vid_w, vid_h, nfrm_div):
    """reference mdl_vog.py:456-463 (on a clone)."""
    p = props5.clone()
    p[..., 0] /= vid_w
    p[..., 1] /= vid_h
    p[..., 2] /= vid_w
    p[..., 3] /= vid_h
    p[..., 4] /= nfrm_div
    return p


def box_bias_head(boxes, w, b):
    """bias[s,p,q] = relu(w . (box[s,p]-box[s,q]) + b) for ONE head
    (compute_pe mdl_vog.py:471-480 + do_cross 'subtract' mdl_srl_utils.py:30-69
    + Linear(5,H)+ReLU mdl_vog.py:446-451). boxes [S,n,5] -> [S,n,n]."""
    d = boxes.unsqueeze(2) - boxes.unsqueeze(1)              # [S,n,n,5]
    return torch.relu(d @ w + b)


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def encoder_layer(x, boxes, nsrl, sd, prefix, pe_name, n_heads, use_rel,
                  quant=None, scope="tx", stash=None, drop=None, site0=0, p_drop=0.0):
    """One (Rel)EncoderLayer (transformer_code.py:84-95,189-203,136-186,
    21-31,73-81). x [S,N,d]; boxes [S,n,5] normalised, N = nsrl*n, token index
    = arg*n + p; the bias on (n x n) is tiled over the nsrl x nsrl arg blocks
    (mdl_vog.py:482-488). scale = sqrt(d_model)."""
    S, N, d = x.shape
    p = f"{prefix}.selfattn.layer."
    # sub-scopes (round 5): a quant hook may treat the pieces of a layer differently ("tx.qk" = the Q.K^T operands ...);
    # hooks that only know "tx" look at scope.split(".")[0]
    xq = _q(quant, scope + ".proj", x)
    q = xq @ _q(quant, scope + ".proj", sd[p + "wq.weight"]).t()
    k = xq @ _q(quant, scope + ".proj", sd[p + "wk.weight"]).t()
    v = xq @ _q(quant, scope + ".proj", sd[p + "wv.weight"]).t()
    scale = math.sqrt(d)
    heads = []
    off = 0
    pm = drop_mask(drop, site0, (S, n_heads, N, N), p_drop) if (drop is not None and p_drop > 0) else None
    for h, dh in enumerate(chunk_sizes(d, n_heads)):
        qh, kh, vh = (t[..., off:off + dh] for t in (q, k, v))
        off += dh
        qh, kh = _q(quant, scope + ".qk", qh, kh)
        logits = qh @ kh.transpose(1, 2)
        if use_rel:
            bh = box_bias_head(boxes, sd[pe_name + ".weight"][h],
                               sd[pe_name + ".bias"][h])
            if nsrl > 1:
                bh = bh.repeat(1, nsrl, nsrl)
            logits = logits + bh
        attn = torch.softmax(logits / scale, dim=-1)
        if pm is not None:
            attn = attn * pm[:, h]                          # self.dropout(F.softmax(...)) (transformer_code.py:50,153)
        attn = _q(quant, scope + ".p", attn)
        vh = _q(quant, scope + ".v", vh)
        heads.append(attn @ vh)
    cat = torch.cat(heads, dim=-1)
    if stash is not None:            # the two inputs of the layer's tail (backward fixtures, oracle/make_golden_bwd.py)
        stash["tail_attn"] = cat
        stash["tail_x"] = x
    a = _q(quant, scope + ".wo", cat) @ _q(quant, scope + ".wo", sd[p + "wo.weight"]).t()
    t = x + _drop(drop, site0 + 1, a, p_drop)               # ResidualBlock: x + dropout(layer(x)) (transformer_code.py:31)
    if stash is not None:
        stash["tail_t"] = t          # its gradient = the gradient of the layer input THROUGH THE TAIL (residual path)
    x1 = layer_norm(t, sd[f"{prefix}.selfattn.layernorm.weight"],
                    sd[f"{prefix}.selfattn.layernorm.bias"])
    f = f"{prefix}.feedforward.layer."
    hdn = torch.relu(_q(quant, scope + ".ffn", x1) @ _q(quant, scope + ".ffn", sd[f + "linear1.weight"]).t()
                     + sd[f + "linear1.bias"])
    y = _q(quant, scope + ".ffn", hdn) @ _q(quant, scope + ".ffn", sd[f + "linear2.weight"]).t() + sd[f + "linear2.bias"]
    y = _drop(drop, site0 + 2, y, p_drop)
    return layer_norm(x1 + y, sd[f"{prefix}.feedforward.layernorm.weight"],
                      sd[f"{prefix}.feedforward.layernorm.bias"])


def transformer(x, boxes, nsrl, sd, name, pe_name, n_layers, n_heads, use_rel,
                quant=None, stash=None, drop=None, stack_id=0, p_drop=0.0):
    """(Rel)Transformer: stack, return last layer output
    (transformer_code.py:227-241,244-279)."""
    for l in range(n_layers):
        x = encoder_layer(x, boxes, nsrl, sd, f"{name}.encoder.layers.{l}",
                          pe_name, n_heads, use_rel, quant, stash=stash if l == n_layers - 1 else None,
                          drop=drop, site0=100 * stack_id + 10 * l, p_drop=p_drop)
    return x


# --------------------------------------------------------------------------- #
# whole forward
# --------------------------------------------------------------------------- #
def _geometry(oc: OracleCfg, ncmp: int):
    """(nc_v, nfrm, nppf) of the model 'video' per conc type
    (mdl_conc_single.py:24-37,131-143; mdl_conc_sep.py:14-26)."""
    if oc.conc_type == "temp":
        return 1, ncmp * oc.nfrm0, oc.nppf0
    if oc.conc_type == "spat":
        return 1, oc.nfrm0, ncmp * oc.nppf0
    return ncmp, oc.nfrm0, oc.nppf0


def forward(oc: OracleCfg, sd: Dict[str, torch.Tensor], inp: Dict[str, torch.Tensor],
            quant: Optional[Callable] = None, keep_stages: bool = False, drop: Optional[int] = None,
            p_obj: float = 0.2, p_mul: float = 0.2):
    """Conc{TEMP,SPAT,SEP}.forward (mdl_conc_single.py:68-127,
    mdl_conc_sep.py:131-217) for ImgGrnd / VidGrnd / VOGNet."""
    st = {}
    words = inp["srl_arg_words_ind"]
    B, nv, nsrl, _ = words.shape
    ncmp = inp["new_srl_idxs"].shape[1]
    sep = oc.conc_type in ("sep", "svsq")
    nc_v, nfrm, nppf = _geometry(oc, ncmp)

    # ---- language (a14-a16)
    tok = srl_arg_seq_to_sent_seq(words, inp["srl_arg_word_mask"], oc.vocab_size)
    lens = inp["srl_arg_word_mask_len"].reshape(B * nv)
    full, hid = lang_encode(tok, lens, sd, oc.rnn_layers, quant, drop=drop)   # drop = seed of the train-mode masks (drop_mask)
    lang = retrieve_srl_args(full, inp["srl_arg_words_capture"],
                             inp["srl_arg_inds_msk"], sd, quant, stash=st if keep_stages else None)   # [B,nv,5,L]
    st.update(tokens=tok, lstm_full_output=full, final_hidden=hid, lang=lang)

    # ---- visual encoders (a13, a12)
    prop = linear(inp["pad_region_feature"].to(F32), sd, "prop_encoder.0", True, quant)
    seg = linear(inp["seg_feature_for_frms"].to(F32), sd, "seg_encoder.0", True, quant)
    st.update(prop_enc=prop, seg_enc=seg)
    if not sep:
        prop = prop.unsqueeze(1)                                  # [B,1,NP,256]
        seg = seg.unsqueeze(1)                                    # [B,1,F,256]
    NP = prop.shape[2]
    Fv = seg.shape[2]
    assert NP == Fv * oc.nppf0 and NP == nfrm * nppf
    seg_b = seg.unsqueeze(3).expand(B, nc_v, Fv, oc.nppf0, seg.shape[-1]).reshape(
        B, nc_v, NP, -1)
    ps = torch.cat([prop, seg_b], dim=-1)                         # [B,nc_v,NP,512]
    st.update(prop_seg=ps)
    props = inp["pad_proposals"].to(F32)
    props5 = (props if sep else props.unsqueeze(1))[..., :5]      # [B,nc_v,NP,5]

    # ---- object transformer (a7, a8)
    if oc.mdl_name in ("vgrnd", "vog") and (oc.mdl_name == "vgrnd" or oc.obj_to_use):
        d = ps.shape[-1]
        if oc.obj_one_frm:
            x = ps.reshape(B * nc_v * nfrm, nppf, d)
            bx = normalise_boxes(props5, oc.vid_w, oc.vid_h, float(nfrm)).reshape(
                B * nc_v * nfrm, nppf, 5)
        else:
            x = ps.reshape(B * nc_v, NP, d)
            bx = normalise_boxes(props5, oc.vid_w, oc.vid_h, 1.0).reshape(B * nc_v, NP, 5)
        obj_stash = {} if keep_stages else None
        x = transformer(x, bx, 1, sd, "obj_txf", "pe_obj_sub_enc.0",
                        oc.obj_layers, oc.obj_heads, oc.obj_use_rel, quant, stash=obj_stash, drop=drop, stack_id=1, p_drop=p_obj)
        if keep_stages:
            st.update(obj_tail_attn=obj_stash["tail_attn"], obj_tail_x=obj_stash["tail_x"], obj_tail_t=obj_stash["tail_t"],
                      obj_boxes=bx, obj_out_seq=x)
        ps = x.reshape(B, nc_v, NP, d)
    st.update(obj_out=ps)

    # ---- vis || lang (a11)
    lang_v = lang.expand(B, nc_v, nsrl, lang.shape[-1]) if lang.shape[1] != nc_v else lang
    conc = torch.cat([
        ps.unsqueeze(2).expand(B, nc_v, nsrl, NP, ps.shape[-1]),
        lang_v.unsqueeze(3).expand(B, nc_v, nsrl, NP, lang.shape[-1])], dim=-1)
    vld = conc.shape[-1]

    # ---- multimodal transformer (a9, a10)
    if oc.mdl_name == "vog":
        x = conc.reshape(B * nc_v, nsrl, nfrm, nppf, vld).transpose(1, 2).reshape(
            B * nc_v * nfrm, nsrl * nppf, vld)
        bx = normalise_boxes(props5, oc.vid_w, oc.vid_h, float(nfrm)).reshape(
            B * nc_v * nfrm, nppf, 5)
        mul_stash = {} if keep_stages else None
        x = transformer(x, bx, nsrl, sd, "mult_txf", "pe_mul_sub_enc.0",
                        oc.mul_layers, oc.mul_heads, oc.mul_use_rel, quant, stash=mul_stash, drop=drop, stack_id=2, p_drop=p_mul)
        if keep_stages:
            st.update(mul_out=x, mul_tail_attn=mul_stash["tail_attn"], mul_tail_x=mul_stash["tail_x"], mul_tail_t=mul_stash["tail_t"],
                      mul_boxes=bx)
        conc = x.reshape(B * nc_v, nfrm, nsrl, nppf, vld).transpose(1, 2).reshape(
            B, nc_v, nsrl, NP, vld)

    # ---- score head (a9 tail / a20)
    h1 = linear(conc, sd, "lin2.0", True, quant, scope="head")
    outs = linear(h1, sd, "lin2.2", False, quant, scope="head").squeeze(-1)   # [B,nc_v,5,NP]

    # ---- masks (a17)
    cm = inp["num_cmp_msk"].to(F32)
    if oc.conc_type == "temp":
        cmsk = cm.view(B, 1, 1, ncmp, 1).expand(B, 1, nsrl, ncmp, oc.nfrm0 * oc.nppf0)
    elif oc.conc_type == "spat":
        cmsk = cm.view(B, 1, 1, 1, ncmp, 1).expand(B, 1, nsrl, oc.nfrm0, ncmp, oc.nppf0)
    else:
        cmsk = cm.view(B, ncmp, 1, 1).expand(B, ncmp, nsrl, NP)
    cmsk = cmsk.reshape(outs.shape)
    am = inp["srl_arg_inds_msk"].to(F32)
    if am.shape[1] != nc_v:
        am = am.expand(B, nc_v, nsrl)
    outs_eval = torch.sigmoid(outs) * am.unsqueeze(-1) * cmsk
    res = {"mdl_outs": outs, "mdl_outs_eval": outs_eval}

    # ---- pred_cmp head (a18)
    if sep:
        seg_mean = seg.mean(dim=-2)                                # [B,ncmp,256]
        verb = hid.reshape(B, nv, -1)
        if verb.shape[1] != ncmp:
            verb = verb.expand(B, ncmp, verb.shape[-1])
        sv = torch.cat([verb, seg_mean], dim=-1)
        vid = linear(linear(sv, sd, "seg_verb_classf.0", True, quant, scope="head"),
                     sd, "seg_verb_classf.2", False, quant, scope="head").squeeze(-1)
        s = torch.sigmoid(outs).max(dim=-1).values                 # [B,ncmp,5]
        vmsk = inp["verb_ind_in_srl"]
        if vmsk.shape[1] != ncmp:
            vmsk = vmsk.expand(B, ncmp)
        s = s.scatter(2, vmsk.unsqueeze(-1), torch.sigmoid(vid).unsqueeze(-1))
        s = s * am
        fin = s.sum(-1) / am.sum(-1) * cm
        res.update(vidf_outs=vid, fin_scores_loss=s * cm.unsqueeze(-1), fin_scores=fin)
    if keep_stages:
        res["stages"] = st
    return res


def pred_head(oc: OracleCfg, out: Dict[str, torch.Tensor], inp: Dict[str, torch.Tensor]):
    """Evaluator{SPAT,TEMP,SEP}.get_out_results_boxes
    (eval_vsrl_corr.py:357-424, 289-345, 162-220)."""
    ev = out["mdl_outs_eval"]
    ncmp = inp["new_srl_idxs"].shape[1]
    B = ev.shape[0]
    nsrl = ev.shape[2]
    nf, np0 = oc.nfrm0, oc.nppf0
    props = inp["pad_proposals"].to(F32)
    if oc.conc_type == "spat":
        r = ev.reshape(B, nsrl, nf, ncmp, np0)
        sc, ix = r.max(dim=-1)                                     # [B,5,nf,ncmp]
        pr = props.reshape(B, 1, nf, ncmp, np0, 7).expand(B, nsrl, nf, ncmp, np0, 7)
        bx = torch.gather(pr, 4, ix[..., None, None].expand(B, nsrl, nf, ncmp, 1, 7)).squeeze(4)
        return {"boxes": bx.transpose(2, 3).contiguous(),
                "scores": sc.transpose(2, 3).contiguous(),
                "indexs": sc.argmax(dim=-1)}
    if oc.conc_type == "temp":
        r = ev.reshape(B, nsrl, ncmp, nf, np0)
        pr = props.reshape(B, 1, ncmp, nf, np0, 7)
    else:
        r = ev.transpose(1, 2).reshape(B, nsrl, ncmp, nf, np0)
        pr = props.reshape(B, 1, ncmp, nf, np0, 7)
    sc, ix = r.max(dim=-1)                                         # [B,5,ncmp,nf]
    pr = pr.expand(B, nsrl, ncmp, nf, np0, 7)
    bx = torch.gather(pr, 4, ix[..., None, None].expand(B, nsrl, ncmp, nf, 1, 7)).squeeze(4)
    if oc.conc_type == "temp":
        idx = torch.zeros(B, nsrl, nf, dtype=F32)
    else:
        idx = out["fin_scores"].argmax(dim=-1).view(B, 1, 1).expand(B, nsrl, nf).contiguous()
    return {"boxes": bx, "scores": sc, "indexs": idx}


def to_torch(d):
    return {k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v
            for k, v in d.items()}


# --------------------------------------------------------------------------- #
# Losses: LossB_TEMP / LossB_SPAT (code/mdl_conc_single.py:180-433),
# LossB_SEP (code/mdl_conc_sep.py:220-447), IoU of utils/box_utils.py:61-118.
# --------------------------------------------------------------------------- #
def bbox_overlaps(props, gt, mask):
    """props [b,N,>=4], gt [b,K,>=4], mask [b,N,K] -> [b,N,K] (box_utils.py:61-118: +1 pixel
    convention, IoU times the mask, 0 where the gt box is empty, -1 where the proposal is empty)."""
    ax = props[:, :, 2] - props[:, :, 0] + 1
    ay = props[:, :, 3] - props[:, :, 1] + 1
    gx = gt[:, :, 2] - gt[:, :, 0] + 1
    gy = gt[:, :, 3] - gt[:, :, 1] + 1
    a_area = (ax * ay).unsqueeze(2)
    g_area = (gx * gy).unsqueeze(1)
    iw = (torch.min(props[:, :, None, 2], gt[:, None, :, 2]) - torch.max(props[:, :, None, 0], gt[:, None, :, 0]) + 1).clamp(min=0)
    ih = (torch.min(props[:, :, None, 3], gt[:, None, :, 3]) - torch.max(props[:, :, None, 1], gt[:, None, :, 1]) + 1).clamp(min=0)
    ov = iw * ih / (a_area + g_area - iw * ih)
    ov = ov * mask.to(ov.dtype)
    ov = ov.masked_fill(((gx == 1) & (gy == 1)).unsqueeze(1), 0.0)
    ov = ov.masked_fill(((ax == 1) & (ay == 1)).unsqueeze(2), -1.0)
    return ov


def _bce_logits(x, t):
    return x.clamp(min=0) - x * t + torch.log1p(torch.exp(-x.abs()))


def loss_forward(oc: "OracleCfg", out, inp, loss_lambda: float = 1.0):
    """-> {'loss', 'mdl_out_loss'[, 'verb_loss']} as the reference LossB_* for oc.conc_type."""
    ct = oc.conc_type
    sep = ct in ("sep", "svsq")
    num_cmp = inp["new_srl_idxs"].shape[1]
    B = inp["target_cmp"].shape[0]
    targ = inp["target_cmp"]
    msk = (inp["pad_frm_mask"].bool() | inp["pad_pnt_mask"].bool().unsqueeze(-1))
    if sep:
        P, G, M = inp["pad_proposals"], inp["pad_gt_bboxs"], msk
        ov = bbox_overlaps(P.flatten(0, 1), G.flatten(0, 1), M.flatten(0, 1)).view(B, num_cmp, P.shape[2], G.shape[2])
        vid = torch.arange(num_cmp).view(1, num_cmp, 1, 1)
        ov_one = ov * (vid == targ.view(B, 1, 1, 1)).to(ov.dtype)
        srl_boxes = inp["srl_boxes"]
        if srl_boxes.shape[1] == 1 and num_cmp > 1:
            srl_boxes = srl_boxes.expand(-1, num_cmp, -1, -1)
        nsrl, nb = srl_boxes.shape[2:]
        NP = ov.shape[2]
        tg = torch.gather(ov_one.unsqueeze(2).expand(B, num_cmp, nsrl, NP, ov.shape[3]), -1,
                          srl_boxes.unsqueeze(3).expand(B, num_cmp, nsrl, NP, nb))
        tg = tg * inp["srl_boxes_lens"].float().unsqueeze(-2)
        targets = tg.max(-1)[0] > 0.5
        tot = _bce_logits(out["mdl_outs"], targets.float())
        abm = inp["srl_arg_boxes_mask"]
        if abm.shape[1] == 1 and num_cmp > 1:
            abm = abm.expand(-1, num_cmp, -1)
        bm = inp["num_cmp_msk"].unsqueeze(-1).expand(*abm.shape).float().unsqueeze(-1).expand(*targets.shape)
        tot = tot * bm
        sel = tot[bm != 0] if abm.max() > 0 else tot
        mdl_out_loss = sel.mean() * tot.shape[-1]
        vl = _bce_logits(out["vidf_outs"], inp["verb_cmp"].float())
        vm = (inp["verb_cross_cmp_msk"].float().sum(-1) > 0).float()
        vl = vl * vm
        verb_loss = vl[vm != 0].mean()
        d = {"loss": mdl_out_loss, "mdl_out_loss": mdl_out_loss, "verb_loss": verb_loss}
        return {k: v * loss_lambda for k, v in d.items()}
    ov = bbox_overlaps(inp["pad_proposals"], inp["pad_gt_bboxs"], msk)           # [B, NPtot, G]
    NPt = ov.shape[1]
    r = torch.arange(NPt)
    if ct == "temp":
        vid = r // (NPt // num_cmp)
    else:
        vid = (r // oc.nppf0) % num_cmp
    ov_one = ov * (vid.view(1, NPt, 1) == targ.view(B, 1, 1)).to(ov.dtype)
    srl_boxes = inp["srl_boxes"]
    nv, nsrl, nb = srl_boxes.shape[1:]
    tg = torch.gather(ov_one.view(B, 1, 1, NPt, -1).expand(B, nv, nsrl, NPt, ov.shape[2]), -1,
                      srl_boxes.unsqueeze(3).expand(B, nv, nsrl, NPt, nb))
    tg = tg * inp["srl_boxes_lens"].float().unsqueeze(-2)
    targets = tg.max(-1)[0] > 0.5
    tot = _bce_logits(out["mdl_outs"], targets.float())
    abm = inp["srl_arg_boxes_mask"]
    bm = abm.float().view(B, nv, nsrl, 1) * inp["num_cmp_msk"].float()[:, vid].view(B, 1, 1, NPt)
    sel = tot[bm != 0] if abm.max() > 0 else tot
    mdl_out_loss = sel.mean() * NPt
    d = {"loss": mdl_out_loss, "mdl_out_loss": mdl_out_loss}
    return {k: v * loss_lambda for k, v in d.items()}


# --------------------------------------------------------------------------- #
# SPAT / TEMP batch assembly: verb_item_getter_SPAT / _TEMP
# (code/dat_loader_simple.py:1046-1338) for a whole batch, numpy. items: leading axes [B, ncmp].
# --------------------------------------------------------------------------- #
def assemble_batch(items, conc_type: str, nfrm0: int, nppf0: int, vid_w: float = 720.0):
    import numpy as np
    P = items["pad_proposals"]
    B, ncmp, NPv, _ = P.shape
    G = items["pad_gt_bboxs"].shape[2]
    vid = np.arange(ncmp, dtype=np.float32).reshape(1, ncmp, 1)
    props, gt = P.copy(), items["pad_gt_bboxs"].copy()
    if conc_type == "spat":        # x += 720 * video (process_props :1081-1103), then (frame, video, prop) order
        for cidx in (0, 2):
            props[..., cidx] = props[..., cidx] + vid * np.float32(vid_w)
            gt[..., cidx] = gt[..., cidx] + vid * np.float32(vid_w)

        def reshuffle(a):           # reshuffle_boxes :1067-1078
            sh = a.shape
            return a.reshape(B, ncmp, nfrm0, nppf0, *sh[3:]).swapaxes(1, 2).reshape(B, ncmp * NPv, *sh[3:])
        segs = items["seg_feature_for_frms"].swapaxes(1, 2)
    else:                           # frame += 10 * video (process_props :1240-1262), plain concatenation
        props[..., 4] = props[..., 4] + vid * np.float32(nfrm0)
        gt[..., 4] = gt[..., 4] + vid * np.float32(nfrm0)

        def reshuffle(a):
            return a.reshape(B, ncmp * NPv, *a.shape[3:])
        segs = items["seg_feature_for_frms"]
    out = {"pad_proposals": reshuffle(props), "pad_region_feature": reshuffle(items["pad_region_feature"]),
           "pad_pnt_mask": reshuffle(items["pad_pnt_mask"]),
           "seg_feature_for_frms": np.ascontiguousarray(segs).reshape(B, ncmp * nfrm0, -1)}
    gts = np.zeros((B, G, 5), np.float32)
    frm = np.ones((B, ncmp * NPv, G), np.uint8)
    srl = items["srl_boxes"].copy()
    nb_tot = items["num_box"].sum(-1)
    for b in range(B):
        rows = [gt[b, v, k] for v in range(ncmp) for k in range(int(items["num_box"][b, v]))]
        if not rows:
            rows = [gt[b, 0, 0]]                              # the reference's fallback (:1106-1108)
        gts[b, :len(rows)] = np.stack(rows)
        nb = int(nb_tot[b])
        frm[b, :, :nb] = (out["pad_proposals"][b, :, 4:5] != gts[b, None, :nb, 4]).astype(np.uint8)
        cum = np.concatenate([[0], np.cumsum(items["num_box"][b])])
        shift = int(cum[int(items["target_cmp"][b])])
        srl[b][items["srl_boxes_lens"][b] > 0] += shift
    out.update({"pad_gt_bboxs": gts, "pad_frm_mask": frm, "srl_boxes": srl, "num_box": nb_tot.astype(np.int64)})
    return out

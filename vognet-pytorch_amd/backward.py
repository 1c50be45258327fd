"""First slice of the training path: the backward of the score head and of the LAST mul_tx
encoder layer's tail on the device (`vog_mul_tail_bwd`, csrc/backward.hip), behind the loss gradient
(`LossB_*.backward` -> `vog_loss_bwd`). What the reference gets from autograd in `Learner.train_epoch`
(utils/trn_utils.py:485-532) for these parameters:

    lin2.{0,2}.{weight,bias}
    mult_txf.encoder.layers.<last>.selfattn.{layer.wo.weight, layernorm.weight, layernorm.bias}
    mult_txf.encoder.layers.<last>.feedforward.{layer.linear1.*, layer.linear2.*, layernorm.*}

plus the gradients of the tail's two inputs (the concatenated attention heads, the layer input through the
residual), where the rest of the backward attaches. fp32; pinned against autograd through the reference
modules (tests/golden/bwd__*.npz).

Second slice: the attention + Q/K/V half of an encoder layer (`vog_attn_f32`: wq / wk / wv, the box-bias
Linear(5, H) `pe_*_sub_enc`, the layer input), which with the tail makes `encoder_layer_backward` - one whole
(Rel)EncoderLayer, with or without the score head behind it.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import lib as L


def tail_param_names(layer: int) -> Dict[str, str]:
    p = f"mult_txf.encoder.layers.{layer}"
    return {"wo": f"{p}.selfattn.layer.wo.weight", "ln1g": f"{p}.selfattn.layernorm.weight",
            "ln1b": f"{p}.selfattn.layernorm.bias", "w1": f"{p}.feedforward.layer.linear1.weight",
            "b1": f"{p}.feedforward.layer.linear1.bias", "w2": f"{p}.feedforward.layer.linear2.weight",
            "b2": f"{p}.feedforward.layer.linear2.bias", "ln2g": f"{p}.feedforward.layernorm.weight",
            "ln2b": f"{p}.feedforward.layernorm.bias", "wl": "lin2.0.weight", "bl": "lin2.0.bias",
            "wl2": "lin2.2.weight", "bl2": "lin2.2.bias"}


def mul_tail_backward(state_dict, layer: int, attn: torch.Tensor, x: torch.Tensor, d_mdl_outs: torch.Tensor,
                      n_vid: int, nfrm: int, nppf: int, nsrl: int, with_input_grads: bool = True) -> Dict[str, torch.Tensor]:
    """-> {reference parameter name: gradient} (+ '_d_attn', '_d_x' [M, d]).

    state_dict: fp32 parameters under the reference's key names (tensors anywhere; copied to the device of
    `attn`); attn / x: [M, d] fp32 device tensors, rows (sequence (video, frame), token arg*nppf + p);
    d_mdl_outs: [n_vid, nsrl, nfrm*nppf] (or the reference's [B, nc_v, nsrl, NP]) as `LossB_*.backward` returns it."""
    lib = L.load()
    dev = attn.device
    assert attn.is_cuda and attn.dtype == torch.float32 and x.shape == attn.shape
    M, d = attn.shape
    assert M == n_vid * nfrm * nsrl * nppf, (M, n_vid, nfrm, nsrl, nppf)
    names = tail_param_names(layer)
    w = {k: state_dict[n].detach().to(dev, torch.float32).contiguous() for k, n in names.items()}
    dh, dhead = w["w1"].shape[0], w["wl"].shape[0]
    assert w["wo"].shape == (d, d) and w["w2"].shape == (d, dh) and w["wl"].shape == (dhead, d)
    g = {k: torch.empty_like(v) for k, v in w.items()}
    d_attn = torch.empty_like(attn) if with_input_grads else None
    d_x = torch.empty_like(attn) if with_input_grads else None
    nb = int(lib.vog_mul_tail_bwd_scratch_bytes(M, d, dh, dhead))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    dmo = d_mdl_outs.to(torch.float32).contiguous()
    assert dmo.numel() == n_vid * nsrl * nfrm * nppf
    a = L.TailBwdArgs()
    keep = [attn.contiguous(), x.contiguous(), dmo, scratch]
    a.attn, a.x, a.d_mdl_outs = L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(dmo)
    a.d_attn = L.ptr(d_attn) if d_attn is not None else None
    a.d_x = L.ptr(d_x) if d_x is not None else None
    for k in ("wo", "ln1g", "ln1b", "w1", "b1", "w2", "b2", "ln2g", "ln2b", "wl", "bl", "wl2"):
        setattr(a, k, L.ptr(w[k]))
    for k in g:
        setattr(a, "g_" + k, L.ptr(g[k]))
    a.scratch, a.scratch_bytes = L.ptr(scratch), nb
    a.M, a.d, a.dh, a.dhead, a.n_vid, a.nfrm, a.nppf, a.nsrl = M, d, dh, dhead, n_vid, nfrm, nppf, nsrl
    L.check(lib.vog_mul_tail_bwd(C.byref(a), L.stream_ptr()), "vog_mul_tail_bwd")
    out = {names[k]: v for k, v in g.items()}
    if with_input_grads:
        out["_d_attn"], out["_d_x"] = d_attn, d_x
    out["_keepalive"] = keep
    return out


def layer_param_names(stack: str, layer: int) -> Dict[str, str]:
    """stack = 'mult_txf' | 'obj_txf' -> the reference's parameter names of one encoder layer."""
    p = f"{stack}.encoder.layers.{layer}"
    return {"wq": f"{p}.selfattn.layer.wq.weight", "wk": f"{p}.selfattn.layer.wk.weight",
            "wv": f"{p}.selfattn.layer.wv.weight", "wo": f"{p}.selfattn.layer.wo.weight",
            "ln1g": f"{p}.selfattn.layernorm.weight", "ln1b": f"{p}.selfattn.layernorm.bias",
            "w1": f"{p}.feedforward.layer.linear1.weight", "b1": f"{p}.feedforward.layer.linear1.bias",
            "w2": f"{p}.feedforward.layer.linear2.weight", "b2": f"{p}.feedforward.layer.linear2.bias",
            "ln2g": f"{p}.feedforward.layernorm.weight", "ln2b": f"{p}.feedforward.layernorm.bias"}


class _Boxes:
    """Proposal rows of the sequences of a layer: props [S*n, stride >= 5] fp32 on the device, normalisers."""

    def __init__(self, props: torch.Tensor, vid_w: float, vid_h: float, nfrm_div: float):
        assert props.is_cuda and props.dtype == torch.float32 and props.dim() == 2 and props.shape[1] >= 5
        self.props, self.vid_w, self.vid_h, self.nfrm_div = props.contiguous(), float(vid_w), float(vid_h), float(nfrm_div)


def _attn_call(w, pe, x, S, N, n, n_heads, boxes, d_cat=None, d_x=None, accumulate_dx=False, want_cat=False, drop=None):
    lib = L.load()
    dev = x.device
    M, d = x.shape
    assert M == S * N and N % n == 0
    nb = int(lib.vog_attn_f32_scratch_bytes(S, N, n, d))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    a = L.AttnF32Args()
    a.x, a.wq, a.wk, a.wv = L.ptr(x), L.ptr(w["wq"]), L.ptr(w["wk"]), L.ptr(w["wv"])
    keep = [x, scratch]
    if boxes is not None:
        a.props, a.prop_stride = L.ptr(boxes.props), boxes.props.shape[1]
        a.vid_w, a.vid_h, a.nfrm_div = boxes.vid_w, boxes.vid_h, boxes.nfrm_div
        a.pe_w, a.pe_b = L.ptr(pe[0]), L.ptr(pe[1])
        assert boxes.props.shape[0] == S * n and pe[0].shape == (n_heads, 5)
    out = {}
    if want_cat or d_cat is None:
        out["cat"] = torch.empty_like(x)
        a.cat_out = L.ptr(out["cat"])
    if d_cat is not None:
        d_cat = d_cat.contiguous()
        keep.append(d_cat)
        a.d_cat = L.ptr(d_cat)
        for k in ("wq", "wk", "wv"):
            out["g_" + k] = torch.empty_like(w[k])
            setattr(a, "g_" + k, L.ptr(out["g_" + k]))
        if boxes is not None:
            out["g_pe_w"], out["g_pe_b"] = torch.empty_like(pe[0]), torch.empty_like(pe[1])
            a.g_pe_w, a.g_pe_b = L.ptr(out["g_pe_w"]), L.ptr(out["g_pe_b"])
        out["d_x"] = d_x if d_x is not None else torch.empty_like(x)
        a.d_x, a.accumulate_dx = L.ptr(out["d_x"]), 1 if (accumulate_dx and d_x is not None) else 0
    a.scratch, a.scratch_bytes = L.ptr(scratch), nb
    a.S, a.N, a.n, a.d, a.n_heads = S, N, n, d, n_heads
    if drop is not None and drop[0] > 0:                   # (p, seed, site of the layer): dropout on the probabilities
        a.drop_p, a.drop_seed, a.drop_site = float(drop[0]), int(drop[1]), int(drop[2])
    L.check(lib.vog_attn_f32(C.byref(a), L.stream_ptr()), "vog_attn_f32")
    out["_keepalive"] = keep
    return out


def _tail_call(w, attn, x, head=None, d_y=None, want_y=False, drop=None):
    """head = (w_head dict {wl, bl, wl2}, d_mdl_outs, n_vid, nfrm, nppf, nsrl) or None (then d_y, or forward only)."""
    lib = L.load()
    dev = x.device
    M, d = x.shape
    dh = w["w1"].shape[0]
    dhead = head[0]["wl"].shape[0] if head is not None else 0
    nb = int(lib.vog_mul_tail_bwd_scratch_bytes(M, d, dh, dhead))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    a = L.TailBwdArgs()
    keep = [attn, x, scratch]
    a.attn, a.x = L.ptr(attn), L.ptr(x)
    for k in ("wo", "ln1g", "ln1b", "w1", "b1", "w2", "b2", "ln2g", "ln2b"):
        setattr(a, k, L.ptr(w[k]))
    out = {}
    bwd = head is not None or d_y is not None
    gkeys = ["wo", "ln1g", "ln1b", "w1", "b1", "w2", "b2", "ln2g", "ln2b"]
    if head is not None:
        wh, dmo, n_vid, nfrm, nppf, nsrl = head
        dmo = dmo.to(torch.float32).contiguous()
        keep.append(dmo)
        a.d_mdl_outs = L.ptr(dmo)
        for k in ("wl", "bl", "wl2"):
            setattr(a, k, L.ptr(wh[k]))
        a.n_vid, a.nfrm, a.nppf, a.nsrl, a.dhead = n_vid, nfrm, nppf, nsrl, dhead
        for k in ("wl", "bl", "wl2", "bl2"):
            out["g_" + k] = torch.empty_like(wh[k])
            setattr(a, "g_" + k, L.ptr(out["g_" + k]))
    else:
        a.no_head = 1
        if d_y is not None:
            d_y = d_y.contiguous()
            keep.append(d_y)
            a.d_y = L.ptr(d_y)
    if bwd:
        for k in gkeys:
            out["g_" + k] = torch.empty_like(w[k])
            setattr(a, "g_" + k, L.ptr(out["g_" + k]))
        out["d_attn"], out["d_x"] = torch.empty_like(x), torch.empty_like(x)
        a.d_attn, a.d_x = L.ptr(out["d_attn"]), L.ptr(out["d_x"])
    if want_y or not bwd:
        out["y"] = torch.empty_like(x)
        a.y_out = L.ptr(out["y"])
    a.scratch, a.scratch_bytes = L.ptr(scratch), nb
    a.M, a.d, a.dh = M, d, dh
    if drop is not None and drop[0] > 0:                   # sites drop_site + 1 / + 2: the two sub-layer outputs
        a.drop_p, a.drop_seed, a.drop_site = float(drop[0]), int(drop[1]), int(drop[2])
    L.check(lib.vog_mul_tail_bwd(C.byref(a), L.stream_ptr()), "vog_mul_tail_bwd")
    out["_keepalive"] = keep
    return out


def _f32(sd, names, dev):
    return {k: sd[n].detach().to(dev, torch.float32).contiguous() for k, n in names.items()}


STACK_ID = {"obj_txf": 1, "mult_txf": 2}      # dropout sites of a layer: 100 * stack + 10 * layer + {0, 1, 2}


def _layer_drop(drop, stack, layer):
    """drop = (p, seed) of a stack or None -> (p, seed, site of this layer)."""
    return None if drop is None or drop[0] <= 0 else (drop[0], drop[1], 100 * STACK_ID[stack] + 10 * layer)


def encoder_layer_forward(state_dict, stack: str, layer: int, pe_name, x: torch.Tensor, S: int, N: int, n: int,
                          n_heads: int, boxes=None, drop=None):
    """fp32 forward of one (Rel)EncoderLayer (the recomputation the backward starts from) -> (y [S*N, d], cat)."""
    dev = x.device
    w = _f32(state_dict, layer_param_names(stack, layer), dev)
    pe = None
    if boxes is not None:
        pe = (state_dict[pe_name + ".weight"].detach().to(dev, torch.float32).contiguous(),
              state_dict[pe_name + ".bias"].detach().to(dev, torch.float32).contiguous())
    x = x.contiguous()
    ld = _layer_drop(drop, stack, layer)
    f = _attn_call(w, pe, x, S, N, n, n_heads, boxes, drop=ld)
    t = _tail_call(w, f["cat"], x, drop=ld)
    return t["y"], f["cat"]


def encoder_layer_backward(state_dict, stack: str, layer: int, pe_name, x: torch.Tensor, S: int, N: int, n: int,
                           n_heads: int, boxes=None, d_y: torch.Tensor = None, head=None, drop=None,
                           cat: torch.Tensor = None) -> Dict[str, torch.Tensor]:
    """Backward of one whole (Rel)EncoderLayer on the device (code/transformer_code.py:128-203).

    x [S*N, d]: the layer's fp32 input. Either `d_y` [S*N, d] (gradient of the layer's output) or `head` =
    (d_mdl_outs, n_vid, nfrm, nppf, nsrl) when the score head `lin2` follows the layer (last mul_tx layer).
    boxes = _Boxes(...) when the layer has the relative-position bias. -> {reference parameter name: gradient,
    '_d_x': gradient of the layer input [S*N, d]}."""
    dev = x.device
    names = layer_param_names(stack, layer)
    w = _f32(state_dict, names, dev)
    pe = None
    if boxes is not None:
        pe = (state_dict[pe_name + ".weight"].detach().to(dev, torch.float32).contiguous(),
              state_dict[pe_name + ".bias"].detach().to(dev, torch.float32).contiguous())
    x = x.contiguous()
    ld = _layer_drop(drop, stack, layer)                                       # train mode: (p, seed) -> the layer's masks
    # the concatenated heads: kept from the forward pass (`cat`) or recomputed
    f = {"cat": cat.contiguous()} if cat is not None else _attn_call(w, pe, x, S, N, n, n_heads, boxes, drop=ld)
    hd = None
    if head is not None:
        hn = {"wl": "lin2.0.weight", "bl": "lin2.0.bias", "wl2": "lin2.2.weight", "bl2": "lin2.2.bias"}
        wh = _f32(state_dict, hn, dev)
        hd = (wh,) + tuple(head)
    t = _tail_call(w, f["cat"], x, head=hd, d_y=d_y, drop=ld)
    b = _attn_call(w, pe, x, S, N, n, n_heads, boxes, d_cat=t["d_attn"], d_x=t["d_x"], accumulate_dx=True, drop=ld)
    out = {names[k]: t["g_" + k] for k in ("wo", "ln1g", "ln1b", "w1", "b1", "w2", "b2", "ln2g", "ln2b")}
    out.update({names[k]: b["g_" + k] for k in ("wq", "wk", "wv")})
    if head is not None:
        out.update({hn[k]: t["g_" + k] for k in hn})
    if boxes is not None:
        out[pe_name + ".weight"], out[pe_name + ".bias"] = b["g_pe_w"], b["g_pe_b"]
    out["_d_x"] = b["d_x"]
    return out


def conc_backward(d_x_mul: torch.Tensor, n_q: int, nc_v: int, nfrm: int, nppf: int, nsrl: int, dobj: int,
                  inds_msk: torch.Tensor = None, lang_per_vid: bool = False):
    """Gradient of mul_tx's input -> (d_ps [(q, v)*NP, dobj], d_lang [(q, v|1)*nsrl, dlang]) (`vog_conc_f32_bwd`)."""
    lib = L.load()
    vld = d_x_mul.shape[-1]
    dlang = vld - dobj
    dev = d_x_mul.device
    d_x_mul = d_x_mul.contiguous()
    assert d_x_mul.numel() == n_q * nc_v * nfrm * nsrl * nppf * vld
    d_ps = torch.empty(n_q * nc_v * nfrm * nppf, dobj, dtype=torch.float32, device=dev)
    d_lang = torch.empty(n_q * (nc_v if lang_per_vid else 1) * nsrl, dlang, dtype=torch.float32, device=dev) if dlang else None
    msk = inds_msk.to(dev, torch.int64).contiguous() if inds_msk is not None else None
    if msk is not None:
        assert msk.numel() == n_q * (nc_v if lang_per_vid else 1) * nsrl
    L.check(lib.vog_conc_f32_bwd(L.ptr(d_x_mul), L.ptr(d_ps), L.ptr(d_lang), L.ptr(msk), n_q, nc_v, nfrm, nppf, nsrl, dobj,
                                 dlang, 1 if lang_per_vid else 0, L.stream_ptr()), "vog_conc_f32_bwd")
    return d_ps, d_lang


def _ptr_view(t: torch.Tensor, col0: int) -> int:
    """Pointer to column col0 of a contiguous 2-D fp32 tensor (a sub-matrix with the parent's row stride)."""
    assert t.is_contiguous() and t.dim() == 2 and t.dtype == torch.float32
    return t.data_ptr() + 4 * col0


def linear_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, relu: bool, dy: torch.Tensor = None, dy_col0: int = 0,
               rep: int = 1, want_dx: bool = False, d_x: torch.Tensor = None, want_y: bool = False):
    """y = act(x W^T + b) and, with dy, its backward (`vog_linear_f32`). dy may be a wider matrix: the gradient of
    this layer's outputs is dy[:, dy_col0 : dy_col0 + N]; rep = downstream replication of the output rows.
    -> dict(y?, g_w, g_b, d_x?)."""
    lib = L.load()
    dev = x.device
    x = x.contiguous()
    M, K = x.shape
    N = w.shape[0]
    assert w.shape == (N, K) and w.is_cuda and x.dtype == torch.float32
    nb = int(lib.vog_linear_f32_scratch_bytes(M, N))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    a = L.LinearF32Args()
    a.x, a.w, a.b, a.relu = L.ptr(x), L.ptr(w), L.ptr(b) if b is not None else None, 1 if relu else 0
    out = {}
    keep = [x, scratch]
    if want_y or dy is None:
        out["y"] = torch.empty(M, N, dtype=torch.float32, device=dev)
        a.y = L.ptr(out["y"])
    if dy is not None:
        assert dy.is_contiguous() and dy.dim() == 2 and dy.shape[0] == M * rep and dy_col0 + N <= dy.shape[1]
        keep.append(dy)
        a.dy, a.ldy, a.rep = _ptr_view(dy, dy_col0), dy.shape[1], rep
        out["g_w"] = torch.empty_like(w)
        a.g_w = L.ptr(out["g_w"])
        if b is not None:
            out["g_b"] = torch.empty_like(b)
            a.g_b = L.ptr(out["g_b"])
        if want_dx or d_x is not None:
            out["d_x"] = d_x if d_x is not None else torch.empty_like(x)
            a.d_x, a.accumulate_dx = L.ptr(out["d_x"]), 1 if d_x is not None else 0
    a.scratch, a.scratch_bytes = L.ptr(scratch), nb
    a.M, a.N, a.K = M, N, K
    L.check(lib.vog_linear_f32(C.byref(a), L.stream_ptr()), "vog_linear_f32")
    out["_keepalive"] = keep
    return out


def stack_forward(state_dict, stack: str, n_layers: int, pe_name, x0: torch.Tensor, S: int, N: int, n: int, n_heads: int,
                  boxes=None, drop=None):
    """fp32 forward of a (Rel)Transformer stack -> (output, [input of every layer], [concatenated heads of every layer]):
    what `stack_backward(..., kept=)` starts from instead of recomputing."""
    xs, cats = [x0.contiguous()], []
    for l in range(n_layers):
        y, cat = encoder_layer_forward(state_dict, stack, l, pe_name, xs[-1], S, N, n, n_heads, boxes, drop=drop)
        cats.append(cat)
        xs.append(y)
    return xs[-1], xs[:-1], cats


def stack_backward(state_dict, stack: str, n_layers: int, pe_name, x0: torch.Tensor, S: int, N: int, n: int, n_heads: int,
                   boxes=None, d_y: torch.Tensor = None, head=None, drop=None, kept=None) -> Dict[str, torch.Tensor]:
    """(Rel)Transformer stack (code/transformer_code.py:227-279): fp32 forward recomputation layer by layer (each
    layer's input kept), then `encoder_layer_backward` from the last layer down. -> {parameter name: gradient,
    '_d_x': gradient of the stack input}. The box-bias Linear is shared by the layers: its gradient is summed."""
    cats = [None] * n_layers
    if kept is not None:                                   # (layer inputs, concatenated heads) of `stack_forward`
        xs, cats = kept
    else:
        xs = [x0.contiguous()]
        for l in range(n_layers - 1):
            y, _ = encoder_layer_forward(state_dict, stack, l, pe_name, xs[-1], S, N, n, n_heads, boxes, drop=drop)
            xs.append(y)
    grads: Dict[str, torch.Tensor] = {}
    d = d_y
    for l in range(n_layers - 1, -1, -1):
        r = encoder_layer_backward(state_dict, stack, l, pe_name, xs[l], S, N, n, n_heads, boxes, d_y=d,
                                   head=head if l == n_layers - 1 else None, drop=drop, cat=cats[l])
        d = r.pop("_d_x")
        for k, v in r.items():
            grads[k] = grads[k] + v if k in grads else v          # (torch add on two gradient tensors: pe_* only)
    grads["_d_x"] = d
    return grads


def score_head_backward(state_dict, x: torch.Tensor, d_mdl_outs: torch.Tensor, n_vid: int, nfrm: int, nppf: int, nsrl: int):
    """lin2 alone (`vog_score_head_f32_bwd`) -> (d_x [M, d], {lin2.* gradients})."""
    lib = L.load()
    dev = x.device
    M, d = x.shape
    hn = {"wl": "lin2.0.weight", "bl": "lin2.0.bias", "wl2": "lin2.2.weight", "bl2": "lin2.2.bias"}
    w = _f32(state_dict, hn, dev)
    dhead = w["wl"].shape[0]
    g = {k: torch.empty_like(v) for k, v in w.items()}
    d_x = torch.empty_like(x)
    nb = int(lib.vog_score_head_f32_bwd_scratch_bytes(M, d, dhead))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    dmo = d_mdl_outs.to(torch.float32).contiguous()
    L.check(lib.vog_score_head_f32_bwd(L.ptr(x.contiguous()), L.ptr(dmo), L.ptr(w["wl"]), L.ptr(w["bl"]), L.ptr(w["wl2"]),
                                       L.ptr(g["wl"]), L.ptr(g["bl"]), L.ptr(g["wl2"]), L.ptr(g["bl2"]), L.ptr(d_x), L.ptr(scratch), nb,
                                       M, d, dhead, n_vid, nfrm, nppf, nsrl, L.stream_ptr()), "vog_score_head_f32_bwd")
    return d_x, {hn[k]: v for k, v in g.items()}


def visual_backward(state_dict, geo: dict, acts: dict, d_mdl_outs: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The visual side of the network behind the loss gradient, on the device in fp32:

        lin2 <- mul_tx <- [obj_tx output | argument vectors] <- obj_tx <- [prop_encoder | seg_encoder]

    geo: B, nc_v, nfrm, nppf, nsrl, nppf0, mul_layers, mul_heads, mul_use_rel, obj_layers, obj_heads, obj_use_rel,
    obj_one_frm, vid_w, vid_h (the model's geometry); acts: 'mul_x' [S*N, vld] (mul_tx's fp32 input), 'obj_x'
    [B*nc_v*NP, dobj] (obj_tx's input = the concatenated encoder outputs), 'prop_feat' [B*nc_v*NP, prop_dim],
    'seg_feat' [B*nc_v*F, seg_dim], 'props' [B*nc_v*NP, >= 5] (pad_proposals), 'inds_msk' [B, nv, nsrl].
    -> gradients by reference parameter name + '_d_lang' (gradient of the masked argument vectors' pre-mask
    activations, where the language side's backward attaches)."""
    g = geo
    B, nc_v, nfrm, nppf, nsrl = g["B"], g["nc_v"], g["nfrm"], g["nppf"], g["nsrl"]
    NP = nfrm * nppf
    out: Dict[str, torch.Tensor] = {}
    props = acts["props"]
    if g["mul_layers"] > 0:
        mb = _Boxes(props, g["vid_w"], g["vid_h"], float(nfrm)) if g["mul_use_rel"] else None
        r = stack_backward(state_dict, "mult_txf", g["mul_layers"], "pe_mul_sub_enc.0", acts["mul_x"], B * nc_v * nfrm, nsrl * nppf,
                           nppf, g["mul_heads"], mb, head=(d_mdl_outs, B * nc_v, nfrm, nppf, nsrl), drop=g.get("drop_mul"),
                           kept=acts.get("mul_kept"))
        d_mul = r.pop("_d_x")
        out.update(r)
    else:
        # ImgGrnd / VidGrnd: lin2 reads the [vis | lang] tokens directly (rows (video, arg, proposal): nfrm = 1, nppf = NP)
        d_mul, hg = score_head_backward(state_dict, acts["mul_x"], d_mdl_outs, B * nc_v, 1, NP, nsrl)
        out.update(hg)
        nfrm, nppf = 1, NP
    dobj = acts["obj_x"].shape[1]
    msk = acts["inds_msk"]
    d_ps, d_lang = conc_backward(d_mul, B, nc_v, nfrm, nppf, nsrl, dobj, inds_msk=msk, lang_per_vid=msk.shape[1] == nc_v and nc_v > 1)
    nfrm, nppf = g["nfrm"], g["nppf"]
    S0 = B * nc_v
    out["_d_lang"] = d_lang
    out["_d_obj_out"] = d_ps
    if g["obj_layers"] > 0:
        if g["obj_one_frm"]:
            S, N, fdiv = S0 * nfrm, nppf, float(nfrm)
        else:
            S, N, fdiv = S0, NP, 1.0
        ob = _Boxes(props, g["vid_w"], g["vid_h"], fdiv) if g["obj_use_rel"] else None
        r = stack_backward(state_dict, "obj_txf", g["obj_layers"], "pe_obj_sub_enc.0", acts["obj_x"], S, N, N, g["obj_heads"], ob,
                           d_y=d_ps, drop=g.get("drop_obj"), kept=acts.get("obj_kept"))
        d_ps = r.pop("_d_x")
        out.update(r)
    out["_d_prop_seg"] = d_ps
    dev = d_ps.device
    wp = state_dict["prop_encoder.0.weight"].detach().to(dev, torch.float32).contiguous()
    bp = state_dict["prop_encoder.0.bias"].detach().to(dev, torch.float32).contiguous()
    ws = state_dict["seg_encoder.0.weight"].detach().to(dev, torch.float32).contiguous()
    bs = state_dict["seg_encoder.0.bias"].detach().to(dev, torch.float32).contiguous()
    penc = wp.shape[0]
    lp = linear_f32(acts["prop_feat"], wp, bp, True, dy=d_ps, dy_col0=0, rep=1)
    ls = linear_f32(acts["seg_feat"], ws, bs, True, dy=d_ps, dy_col0=penc, rep=g["nppf0"])
    out["prop_encoder.0.weight"], out["prop_encoder.0.bias"] = lp["g_w"], lp["g_b"]
    out["seg_encoder.0.weight"], out["seg_encoder.0.bias"] = ls["g_w"], ls["g_b"]
    return out


def lang_param_names(layers: int) -> Dict[str, str]:
    n = {"emb": "lstm_encoder.embed_tokens.weight", "w_proj": "lstm_out_feat_proj.0.weight", "b_proj": "lstm_out_feat_proj.0.bias",
         "w_arg": "srl_arg_words_out_enc.0.weight", "b_arg": "srl_arg_words_out_enc.0.bias"}
    for l in range(layers):
        for dr, sfx in enumerate(("", "_reverse")):
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                n[f"{k}:{l}:{dr}"] = f"lstm_encoder.lstm.{k}_l{l}{sfx}"
    return n


def language_backward(state_dict, batch: dict, T: int, layers: int, d_lang_enc: torch.Tensor = None, drop=None,
                      forward_scratch: torch.Tensor = None) -> Dict[str, torch.Tensor]:
    """The language side on the device in fp32 (`vog_lang_f32`): embedding, packed BiLSTM (back-propagation through
    time), lstm_out_feat_proj, srl_arg_words_out_enc. batch: the model's input dict (device int64 tensors
    srl_arg_words_ind [B, nv, nsrl, sl], srl_arg_word_mask [B, nv, ml], srl_arg_word_mask_len [B, nv],
    srl_arg_words_capture [B, nv, nsrl, 2]). d_lang_enc [B*nv*nsrl, L] (`visual_backward`'s '_d_lang') or None for
    the forward only. -> {parameter name: gradient} (+ '_lang_enc', '_full' forward activations)."""
    lib = L.load()
    words = batch["srl_arg_words_ind"]
    dev = words.device
    B, nv, nsrl, sl = words.shape
    Bn = B * nv
    names = lang_param_names(layers)
    w = {k: state_dict[n].detach().to(dev, torch.float32).contiguous() for k, n in names.items()}
    E, R = w["emb"].shape[1], w["weight_hh:0:0"].shape[1]
    D, Lo = w["w_proj"].shape[0], w["w_arg"].shape[0]
    a = L.LangF32Args()
    ints = [words.reshape(Bn, nsrl * sl).to(torch.int64).contiguous(),
            batch["srl_arg_word_mask"].reshape(Bn, -1).to(torch.int64).contiguous(),
            batch["srl_arg_word_mask_len"].reshape(Bn).to(torch.int64).contiguous(),
            batch["srl_arg_words_capture"].reshape(Bn, nsrl, 2).to(torch.int64).contiguous()]
    a.words_ind, a.word_mask, a.lens, a.capture = (L.ptr(t) for t in ints)
    a.Bn, a.nsrl, a.words_len, a.mask_len, a.T = Bn, nsrl, nsrl * sl, ints[1].shape[1], T
    a.vocab_size, a.E, a.R, a.layers, a.D, a.L = w["emb"].shape[0] - 1, E, R, layers, D, Lo
    a.emb = L.ptr(w["emb"])
    for l in range(layers):
        for dr in range(2):
            a.w_ih[l][dr], a.w_hh[l][dr] = L.ptr(w[f"weight_ih:{l}:{dr}"]), L.ptr(w[f"weight_hh:{l}:{dr}"])
            a.b_ih[l][dr], a.b_hh[l][dr] = L.ptr(w[f"bias_ih:{l}:{dr}"]), L.ptr(w[f"bias_hh:{l}:{dr}"])
    a.w_proj, a.b_proj, a.w_arg, a.b_arg = L.ptr(w["w_proj"]), L.ptr(w["b_proj"]), L.ptr(w["w_arg"]), L.ptr(w["b_arg"])
    out = {"_lang_enc": torch.empty(Bn * nsrl, Lo, dtype=torch.float32, device=dev),
           "_full": torch.empty(Bn * T, D, dtype=torch.float32, device=dev),
           "_hid": torch.empty(Bn, D, dtype=torch.float32, device=dev)}
    a.lang_enc_out, a.full_out, a.hid_out = L.ptr(out["_lang_enc"]), L.ptr(out["_full"]), L.ptr(out["_hid"])
    g = {}
    if d_lang_enc is not None:
        d_lang_enc = d_lang_enc.to(torch.float32).contiguous()
        assert d_lang_enc.shape == (Bn * nsrl, Lo)
        a.d_lang_enc = L.ptr(d_lang_enc)
        g = {k: torch.empty_like(v) for k, v in w.items()}
        a.g_emb, a.g_w_proj, a.g_b_proj, a.g_w_arg, a.g_b_arg = (L.ptr(g[k]) for k in ("emb", "w_proj", "b_proj", "w_arg", "b_arg"))
        for l in range(layers):
            for dr in range(2):
                a.g_w_ih[l][dr], a.g_w_hh[l][dr] = L.ptr(g[f"weight_ih:{l}:{dr}"]), L.ptr(g[f"weight_hh:{l}:{dr}"])
                a.g_b_ih[l][dr], a.g_b_hh[l][dr] = L.ptr(g[f"bias_ih:{l}:{dr}"]), L.ptr(g[f"bias_hh:{l}:{dr}"])
    if drop is not None:                                   # train mode: (p_in, p_out, seed) of LSTMEncoder's dropouts
        a.drop_in, a.drop_out, a.drop_seed = float(drop[0]), float(drop[1]), int(drop[2])
    nb = int(lib.vog_lang_f32_scratch_bytes(Bn, T, nsrl, E, R, layers, D, Lo))
    if forward_scratch is not None:                        # '_scratch' of the forward-only call on the same batch / weights / seed
        assert d_lang_enc is not None and forward_scratch.numel() == nb
        scratch, a.reuse_forward = forward_scratch, 1
    else:
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    a.scratch, a.scratch_bytes = L.ptr(scratch), nb
    L.check(lib.vog_lang_f32(C.byref(a), L.stream_ptr()), "vog_lang_f32")
    out["_scratch"] = scratch
    out.update({names[k]: v for k, v in g.items()})
    out["_keepalive"] = [ints, w, scratch, d_lang_enc]
    return out

"""First slice of the training path: the backward of the score head and of the LAST mul_tx
encoder layer's tail on the device (`vog_mul_tail_bwd`, csrc/backward.hip), behind the loss gradient
(`LossB_*.backward` -> `vog_loss_bwd`). What the reference gets from autograd in `Learner.train_epoch`
(utils/trn_utils.py:485-532) for these parameters:

    lin2.{0,2}.{weight,bias}
    mult_txf.encoder.layers.<last>.selfattn.{layer.wo.weight, layernorm.weight, layernorm.bias}
    mult_txf.encoder.layers.<last>.feedforward.{layer.linear1.*, layer.linear2.*, layernorm.*}

plus the gradients of the tail's two inputs (the concatenated attention heads, the layer input through the
residual), where the rest of the backward (attention, QKV, encoders, BiLSTM) will attach. fp32; pinned
against autograd through the reference modules (tests/golden/bwd__*.npz). No optimizer and no gradient
all-reduce yet.
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

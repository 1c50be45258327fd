"""Concat-strategy mixins (reference code/mdl_conc_single.py:23-177,
code/mdl_conc_sep.py:13-217 — forward halves only). How the (video, frame,
proposal) axes fold into (sequence, token) axes per strategy is implemented in
csrc/forward.hip `make_geo`; the mixins pin `cfg.ds.conc_type` to the class and
carry the loss classes' names for the selector.
"""
from __future__ import annotations

import torch
from torch import nn


class ConcBase:
    conc_types = ()

    def set_args_conc(self):
        ct = self.cfg.ds.conc_type
        assert ct in self.conc_types, f"cfg.ds.conc_type={ct!r} not served by {type(self).__name__}"


class ConcTEMP(ConcBase):
    """4 videos concatenated in time: 1 model video, nfrm = ncmp*10, nppf = nppf0."""
    conc_types = ("temp",)


class ConcSPAT(ConcBase):
    """4 videos tiled in space: 1 model video, nfrm = 10, nppf = ncmp*nppf0."""
    conc_types = ("spat",)


class ConcSEP(ConcBase):
    """videos kept separate (+ pred_cmp head); svsq = single video."""
    conc_types = ("sep", "svsq")

    def set_args_conc(self):
        ConcBase.set_args_conc(self)
        self.nfrms = self.num_sampled_frm
        self.nppf = self.num_prop_per_frm


class _LossB(nn.Module):
    """`loss_fn(out, inp) -> {'loss', 'mdl_out_loss'[, 'verb_loss']}` of the reference
    (LossB_TEMP / LossB_SPAT code/mdl_conc_single.py:180-433, LossB_SEP code/mdl_conc_sep.py:220-447;
    the evaluator calls it for every validation batch, code/eval_vsrl_corr.py:119). The IoU targets
    (utils/box_utils.py:61-118), the target selection and the masked BCE run on the device in
    `vog_loss_fwd` (csrc/loss.hip); the returned values are 0-dim device tensors (no sync here).
    `backward(loss_dict)` (the dict `forward` returned, or any dict after it: the last call's context is kept
    on the module) returns d loss / d mdl_outs (`vog_loss_bwd`): the first link of the training path
    (SURVEY.md 8(f) rank 4); nothing behind it (score head, transformers, BiLSTM) has a backward yet."""
    loss_keys = ["loss", "mdl_out_loss"]
    conc_types = ()

    def __init__(self, cfg, comm):
        super().__init__()
        self.cfg, self.comm = cfg, comm
        self.loss_lambda = float(cfg.loss.loss_lambda)
        ct = cfg.ds.conc_type
        assert ct in self.conc_types, f"cfg.ds.conc_type={ct!r} not served by {type(self).__name__}"
        self.nppf0 = int(comm["num_prop_per_frm"])

    def forward(self, out, inp):
        import ctypes as C
        from . import lib as L
        lib = L.load()
        mo = out["mdl_outs"]
        assert mo.is_cuda and mo.dtype == torch.float32, "the loss runs on the device outputs of the forward"
        sep = self.cfg.ds.conc_type in ("sep", "svsq")
        B, nvo, nsrl, NP = mo.shape
        ncmp = inp["num_cmp_msk"].shape[1]

        def dev(k, dt):
            t = inp[k]
            if t.dtype == torch.bool:
                t = t.to(torch.uint8)
            assert t.is_cuda and t.dtype == dt, (k, t.dtype, dt)
            return t.contiguous()

        keep = [mo.contiguous()]
        a = L.LossArgs()
        a.mdl_outs = L.ptr(keep[0])
        for k, dt in (("pad_proposals", torch.float32), ("pad_gt_bboxs", torch.float32), ("pad_frm_mask", torch.uint8),
                      ("pad_pnt_mask", torch.uint8), ("srl_boxes", torch.int64), ("srl_boxes_lens", torch.int64),
                      ("srl_arg_boxes_mask", torch.int64), ("target_cmp", torch.int64), ("num_cmp_msk", torch.int64)):
            t = dev(k, dt)
            keep.append(t)
            setattr(a, k, L.ptr(t))
        if sep:
            for k in ("verb_cmp", "verb_cross_cmp_msk"):
                t = dev(k, torch.int64)
                keep.append(t)
                setattr(a, k, L.ptr(t))
            vo = out["vidf_outs"].contiguous()
            keep.append(vo)
            a.vidf_outs = L.ptr(vo)
        a.B, a.ncmp, a.nv, a.nsrl = B, ncmp, inp["srl_boxes"].shape[1], nsrl
        a.nbox, a.NP, a.G, a.nppf0 = inp["srl_boxes"].shape[3], NP, inp["pad_gt_bboxs"].shape[-2], self.nppf0
        a.conc_type, a.loss_lambda = L.CONC_TYPE[self.cfg.ds.conc_type], self.loss_lambda
        assert inp["pad_proposals"].shape[-2] == NP and inp["pad_gt_bboxs"].shape[-1] == 5
        res = torch.empty(6, dtype=torch.float32, device=mo.device)
        scr = torch.empty(max(16, int(lib.vog_loss_scratch_bytes(C.byref(a)))), dtype=torch.uint8, device=mo.device)
        a.out, a.scratch = L.ptr(res), L.ptr(scr)
        L.check(lib.vog_loss_fwd(C.byref(a), L.stream_ptr()), "vog_loss_fwd")
        d = {"loss": res[0], "mdl_out_loss": res[1]}
        if sep:
            d["verb_loss"] = res[2]
        # exactly the `loss_keys` tensors, as the reference (consumers iterate the dict and call .detach() /
        # reduce_dict on every value, code/eval_vsrl_corr.py:119-122). What `backward` needs - the argument
        # block and the tensors it points into - stays on the module and rides on the `loss` tensor.
        self._last = (a, keep, scr)
        d["loss"]._vog_loss_ctx = self._last
        return d

    def backward(self, loss_dict, with_verb: bool = False):
        """d loss / d mdl_outs (and, `with_verb`, d verb_loss / d vidf_outs) for the dict `forward` returned:
        `vog_loss_bwd`, the first link of the training path (the reference gets it from autograd)."""
        import ctypes as C
        from . import lib as L
        lib = L.load()
        ctx = getattr(loss_dict["loss"], "_vog_loss_ctx", None) or self._last
        a, keep = ctx[0], ctx[1]
        mo = keep[0]
        g = torch.empty_like(mo)
        gv = None
        if with_verb:
            assert self.cfg.ds.conc_type in ("sep", "svsq")
            gv = torch.empty((a.B, a.ncmp), dtype=torch.float32, device=mo.device)
        L.check(lib.vog_loss_bwd(C.byref(a), L.ptr(g), L.ptr(gv) if gv is not None else None, L.stream_ptr()),
                "vog_loss_bwd")
        return (g, gv) if with_verb else g


class LossB_TEMP(_LossB):
    conc_types = ("temp",)


class LossB_SPAT(_LossB):
    conc_types = ("spat",)


class LossB_SEP(_LossB):
    loss_keys = ["loss", "mdl_out_loss", "verb_loss"]
    conc_types = ("sep", "svsq")

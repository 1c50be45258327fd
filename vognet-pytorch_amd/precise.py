"""The fp32 forward as the product path's PRECISE mode (round 5).

The 16-bit forward holds the 1e-3 bound on pred_scores only while the attention logits are not too sharp: the softmax
of code/transformer_code.py:141-155 turns an absolute logit error into a relative probability error, and the logit error
of 16-bit operands grows with the logit scale. `engine.attention_sharpness` measures that scale from the weights
(||Wq_h^T Wk_h||_F / sqrt(d), times the mean square of the layer's input); the envelope per operand type is written in
DESIGN.md section 2. Beyond the f16 envelope `cfg.hip.tx_dtype = auto` routes every forward through the fp32 kernels of
csrc/backward.hip instead - the forward half of the training path (`train.FP32Trainer.forward`: fp32 MFMA GEMMs, row-wise
softmax, step-by-step BiLSTM; pinned against the reference goldens at 1e-6) - followed by the same exact heads as the
16-bit path (`vog_score_head` masks, `vog_pred_cmp_head`, `vog_pred_head`). ~10x slower than the f16 forward (2.2 ms per
cfg-2 batch) and still 25x the CPU reference; it exists so that a sharply trained checkpoint
(EXPTS.md:95-189, utils/trn_utils.py:534-614) never silently leaves the bound.

Everything is a C-ABI call into libvog_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import lib as L


class PreciseForward:
    def __init__(self, eng, state_dict):
        from .train import FP32Trainer
        self.eng = eng
        comm = {"vocab_size": eng.desc.vocab_size, "num_prop_per_frm": eng.desc.nppf0}
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in state_dict.items()}
        # reference key styles (`module.` prefix of DistributedDataParallel checkpoints, utils/trn_utils.py:560-575)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.tr = FP32Trainer(eng.cfg, comm, sd, loss_fn=None, lr=0.0, device=str(eng.device))
        dev = eng.device
        self._one = torch.ones(1, dtype=torch.float32, device=dev)
        self._zero = torch.zeros(1, dtype=torch.float32, device=dev)

    def run(self, inp: Dict[str, torch.Tensor], out: Dict[str, torch.Tensor], T=None) -> None:
        """fp32 forward of `inp` on the CURRENT stream; overwrites mdl_outs / mdl_outs_eval (/ vidf_outs, fin_scores*,
        pred_rec) of `out` in place. `T`: the longest sentence when the caller knows it (a slot's launch stays
        asynchronous - no host read of the lengths on the launch stream)."""
        eng, lib, d = self.eng, self.eng.lib, self.eng.desc
        o, acts, g = self.tr.forward(inp, T=T)
        st = L.stream_ptr()
        B, nc_v, nsrl = g["B"], g["nc_v"], g["nsrl"]
        NP = g["nfrm"] * g["nppf"]
        ncmp = inp["num_cmp_msk"].shape[1]
        nvl = inp["srl_arg_inds_msk"].shape[1]
        logits = o["mdl_outs"].contiguous()
        arg_msk = inp["srl_arg_inds_msk"].contiguous()
        cmp_msk = inp["num_cmp_msk"].contiguous()
        # sigmoid * masks (mdl_conc_single.py:118-122): vog_score_head with the logits as a 1-wide hidden layer (x * 1 + 0)
        a = L.ScoreArgs()
        a.h1, a.w2, a.b2 = L.ptr(logits), L.ptr(self._one), L.ptr(self._zero)
        a.arg_msk, a.cmp_msk = L.ptr(arg_msk), L.ptr(cmp_msk)
        a.outs, a.outs_eval = L.ptr(out["mdl_outs"]), L.ptr(out["mdl_outs_eval"])
        a.n_vid, a.nfrm, a.nppf, a.nsrl, a.dh = B * nc_v, 1, NP, nsrl, 1
        a.conc_type, a.ncmp, a.nc_v, a.nvl = d.conc_type, ncmp, nc_v, nvl
        a.nfrm0, a.nppf0 = d.nfrm0, d.nppf0
        L.check(lib.vog_score_head(C.byref(a), st), "vog_score_head")
        keep = [logits, arg_msk, cmp_msk, acts]
        if eng.sep:
            p = self.tr.params
            pc = L.PredcmpArgs()
            hid = acts["hid"].contiguous()
            ps = acts["obj_x"]
            pc.final_hidden, pc.prop_seg = L.ptr(hid), L.ptr(ps)
            pc.w0, pc.b0 = L.ptr(p["seg_verb_classf.0.weight"]), L.ptr(p["seg_verb_classf.0.bias"])
            pc.w2, pc.b2 = L.ptr(p["seg_verb_classf.2.weight"]), L.ptr(p["seg_verb_classf.2.bias"])
            pc.outs, pc.arg_msk, pc.cmp_msk = L.ptr(out["mdl_outs"]), L.ptr(arg_msk), L.ptr(cmp_msk)
            pc.verb_ind = L.ptr(out["_verb"])
            pc.vidf_outs, pc.fin_scores_loss, pc.fin_scores = (L.ptr(out["vidf_outs"]), L.ptr(out["fin_scores_loss"]),
                                                               L.ptr(out["fin_scores"]))
            pc.B, pc.ncmp, pc.nvl, pc.nsrl, pc.NP = B, ncmp, nvl, nsrl, NP
            pc.nfrm0, pc.nppf0, pc.L = d.nfrm0, d.nppf0, hid.shape[1]
            pc.dp0, pc.dps = d.prop_enc, ps.shape[1]
            L.check(lib.vog_pred_cmp_head(C.byref(pc), st), "vog_pred_cmp_head")
            keep.append(hid)
        if out.get("pred_rec") is not None:
            pa = L.PredArgs()
            props = inp["pad_proposals"].contiguous()
            pa.outs_eval, pa.props = L.ptr(out["mdl_outs_eval"]), L.ptr(props)
            pa.fin_scores = L.ptr(out["fin_scores"]) if eng.sep else None
            pa.rec = L.ptr(out["pred_rec"])
            pa.B, pa.ncmp, pa.nsrl, pa.nfrm0, pa.nppf0, pa.conc_type = B, ncmp, nsrl, d.nfrm0, d.nppf0, d.conc_type
            L.check(lib.vog_pred_head(C.byref(pa), st), "vog_pred_head")
            keep.append(props)
        out["_precise_keepalive"] = keep

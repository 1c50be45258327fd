"""Prediction head of the "evaluator" on the plugin surface.

Same classes / ctor / `get_out_results_boxes` / record format as reference
code/eval_vsrl_corr.py (Evaluator{SEP,TEMP,SPAT}: 24-33, 162-424), with the
arg-max + box gather done by `vog_pred_head` on the GPU and the cross-rank
gather done by one RCCL all-gather (dist.py) instead of pickle files.
Metrics (`GroundEval_*`, code/eval_fn_corr.py) need the dataset annotations and
are out of scope: attach one via `self.grnd_eval`.
"""
from __future__ import annotations

import ctypes as C
import pickle
from pathlib import Path

import torch

from . import dist as D
from . import lib as L


class Evaluator(torch.nn.Module):
    conc_type = None

    def __init__(self, cfg, comm, device):
        super().__init__()
        self.cfg = cfg
        self.comm = comm
        self.met_keys = ["avg1", "macro_avg1"]
        self.num_prop_per_frm = comm["num_prop_per_frm"]
        self.num_frms = cfg.ds.num_sampled_frm
        self.num_props = self.num_prop_per_frm * self.num_frms
        self.device = device
        self.grnd_eval = None
        self.after_init()

    def after_init(self):
        self.met_keys = ["avg1", "avg1_cons", "avg1_vidf", "avg1_strict"]
        self.num_sampled_frm = self.num_frms

    # ---- device head -----------------------------------------------------------
    def _records(self, out, inp):
        if "_pred_rec" in out:
            return out["_pred_rec"]
        lib = L.load()
        ev = out["mdl_outs_eval"].contiguous()
        B = ev.shape[0]
        ncmp = inp["new_srl_idxs"].size(1)
        nsrl = ev.shape[2]
        rb = int(lib.vog_pred_record_bytes(ncmp, nsrl, self.num_frms))
        rec = torch.empty(B, rb // 4, dtype=torch.float32, device=ev.device)
        a = L.PredArgs()
        a.outs_eval = L.ptr(ev)
        a.props = L.ptr(inp["pad_proposals"].contiguous())
        a.fin_scores = L.ptr(out["fin_scores"].contiguous()) if "fin_scores" in out else None
        a.rec = L.ptr(rec)
        a.B, a.ncmp, a.nsrl, a.nfrm0, a.nppf0 = B, ncmp, nsrl, self.num_frms, self.num_prop_per_frm
        a.conc_type = L.CONC_TYPE[self.conc_type]
        L.check(lib.vog_pred_head(C.byref(a), L.stream_ptr()), "vog_pred_head")
        return rec

    def unpack(self, rec, ncmp, nsrl):
        B = rec.shape[0]
        nb = nsrl * ncmp * self.num_frms
        boxes = rec[:, : nb * 7].reshape(B, nsrl, ncmp, self.num_frms, 7)
        scores = rec[:, nb * 7: nb * 8].reshape(B, nsrl, ncmp, self.num_frms)
        if self.conc_type == "temp":     # reference returns float zeros (eval_vsrl_corr.py:338-340)
            idx = torch.zeros(B, nsrl, self.num_frms, dtype=torch.float32, device=rec.device)
        else:
            idx = rec[:, nb * 8:].contiguous().view(torch.int64).reshape(B, nsrl, self.num_frms)
        return {"boxes": boxes, "scores": scores, "indexs": idx}

    def get_out_results_boxes(self, out_result_dict, inp):
        """-> {'boxes' [B,nsrl,ncmp,nfrm,7], 'scores' [B,nsrl,ncmp,nfrm], 'indexs' [B,nsrl,nfrm]}"""
        assert isinstance(out_result_dict, dict)
        rec = self._records(out_result_dict, inp)
        ncmp = inp["new_srl_idxs"].size(1)
        nsrl = out_result_dict["mdl_outs_eval"].shape[2]
        return self.unpack(rec, ncmp, nsrl)

    def forward_one_batch(self, out_result, inp):
        """Python-list records in the reference's format (eval_vsrl_corr.py:247-273)."""
        r = self.get_out_results_boxes(out_result, inp)
        cols = {
            "pred_boxes": r["boxes"], "pred_scores": r["scores"], "pred_cmp": r["indexs"],
            "idx_vid": inp["ann_idx"], "idx_verbs": inp["new_srl_idxs"], "idx_sent": inp["sent_idx"],
            "cmp_msk": inp["num_cmp_msk"], "targ_cmp": inp["target_cmp"], "perm": inp["permute"],
            "perm_inv": inp["permute_inv"],
        }
        cols = {k: v.detach().cpu().tolist() for k, v in cols.items()}
        n = len(cols["pred_boxes"])
        return [{k: v[i] for k, v in cols.items()} for i in range(n)]

    def forward(self, model, loss_fn, dl, dl_name, rank=0, pred_path=None, mb=None):
        """Loop a dataloader, collect prediction records; ranks exchange the
        packed device records with ONE all-gather per batch; rank 0 writes the
        merged pickle in the reference format. Loss / metric values are only
        produced when the (out-of-scope) loss_fn / grnd_eval are supplied."""
        model.eval()
        results = []
        meta_keys = ("ann_idx", "new_srl_idxs", "sent_idx", "num_cmp_msk", "target_cmp",
                     "permute", "permute_inv")
        for batch in dl:
            batch = {k: v.to(self.device) for k, v in batch.items()}
            with torch.no_grad():
                out = model(batch)
            rec = D.all_gather_records(self._records(out, batch))
            ncmp = batch["new_srl_idxs"].size(1)
            nsrl = out["mdl_outs_eval"].shape[2]
            r = self.unpack(rec, ncmp, nsrl)
            meta = {k: D.all_gather_records(batch[k]) for k in meta_keys if k in batch}
            if D.is_main_process():
                cols = {"pred_boxes": r["boxes"], "pred_scores": r["scores"], "pred_cmp": r["indexs"]}
                names = {"ann_idx": "idx_vid", "new_srl_idxs": "idx_verbs", "sent_idx": "idx_sent",
                         "num_cmp_msk": "cmp_msk", "target_cmp": "targ_cmp", "permute": "perm",
                         "permute_inv": "perm_inv"}
                cols.update({names[k]: v for k, v in meta.items()})
                cols = {k: v.detach().cpu().tolist() for k, v in cols.items()}
                n = len(cols["pred_boxes"])
                results += [{k: v[i] for k, v in cols.items()} for i in range(n)]
        val_acc = {k: torch.tensor(0.0) for k in self.met_keys}
        if D.is_main_process() and pred_path is not None:
            fname = Path(pred_path) / f"{dl_name}_{rank}.pkl"
            fname.parent.mkdir(parents=True, exist_ok=True)
            with open(fname, "wb") as f:
                pickle.dump(results, f)
            if self.grnd_eval is not None:
                acc = self.grnd_eval.eval_ground_acc(fname)
                val_acc = {k: torch.tensor(v) for k, v in acc.items() if k in self.met_keys}
        D.synchronize()
        return {}, val_acc


class EvaluatorSEP(Evaluator):
    conc_type = "sep"


class EvaluatorTEMP(Evaluator):
    conc_type = "temp"


class EvaluatorSPAT(Evaluator):
    conc_type = "spat"

"""Prediction head of the "evaluator" on the plugin surface.

Same classes / ctor / `get_out_results_boxes` / record format as reference
code/eval_vsrl_corr.py (Evaluator{SEP,TEMP,SPAT}: 24-33, 162-424), with the
arg-max + box gather done by `vog_pred_head` on the GPU and the cross-rank
gather done by one RCCL all-gather (dist.py) instead of pickle files.
The validation loss comes from the device loss (`mdl_conc.LossB_*` -> `vog_loss_fwd`). Metrics: when the
annotation files of `cfg.ds` (`val_ds4_inds`, `anet_ent_annot_file`) exist, `after_init` builds the
`GroundEval_*` of the concatenation type (eval_fn_corr.py in this package, pinned against the reference's)
as the reference's `after_init` does (eval_vsrl_corr.py:154-158, 277-283, 349-351) and rank 0 scores the
merged pickle at the end of `forward`; without the files `val_acc` is zeros.
"""
from __future__ import annotations

import ctypes as C
import pickle
from pathlib import Path

import numpy as np
import torch

from . import dist as D
from . import fast_pickle
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
        self.grnd_eval = self._make_grnd_eval()

    def _make_grnd_eval(self):
        import os
        from . import eval_fn_corr as M
        ds = self.cfg.ds
        files = [getattr(ds, k, None) if not isinstance(ds, dict) else ds.get(k) for k in ("val_ds4_inds", "anet_ent_annot_file")]
        if self.conc_type is None or not all(isinstance(f, str) and os.path.isfile(f) for f in files):
            return None
        cls = {"sep": M.GroundEval_SEP, "temp": M.GroundEval_TEMP, "spat": M.GroundEval_SPAT}[self.conc_type]
        return cls(self.cfg, self.comm)

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

    META_KEYS = ("ann_idx", "new_srl_idxs", "sent_idx", "num_cmp_msk", "target_cmp", "permute", "permute_inv")
    META_NAMES = {"ann_idx": "idx_vid", "new_srl_idxs": "idx_verbs", "sent_idx": "idx_sent", "num_cmp_msk": "cmp_msk",
                  "target_cmp": "targ_cmp", "permute": "perm", "permute_inv": "perm_inv"}
    GATHER_EVERY = 16          # batches per cross-rank exchange (dist.RecordRing half)

    def forward(self, model, loss_fn, dl, dl_name, rank=0, pred_path=None, mb=None):
        """The validation loop of the reference (code/eval_vsrl_corr.py:101-150): forward, loss_fn(out,
        batch), prediction records; rank 0 writes `<pred_path>/<dl_name>_<rank>.pkl` in the reference
        record format (:247-273) and returns (loss dict, metric dict).

        Exchange: the reference pickles per-rank predictions and rank 0 re-reads the files (:125-140).
        Here every batch's packed device records travel through ONE all-gather per GATHER_EVERY batches
        (dist.RecordRing; a collective per batch costs 12-20 us per step) and the batches' metadata (ids,
        masks, permutations: int64, known on the HOST before the batch is uploaded) stays on the host and
        is exchanged once, at the end. Record order on rank 0 = (rank, batch, query): all of rank 0's
        records, then all of rank 1's, ... - the order in which the reference appends the per-rank files
        (:131-137).

        Host side of the loop: the batches are uploaded `depth` batches ahead on a copy stream
        (dat_loader_simple.DevicePrefetcher) and the longest sentence is taken from the host copy of the
        lengths, so that no step of the loop waits for the device (the per-key blocking `.to(device)` and the
        `.item()` of the reference's loop cost 1.5 ms per cfg-2 batch)."""
        from .dat_loader_simple import DevicePrefetcher
        model.eval()
        world = D.get_world_size()
        rec_rows = [[] for _ in range(world)]          # rank 0: per rank, the gathered record rows of every half (numpy)
        meta_rows = []                                 # this rank: per ring entry, int64 [rows_ring, W + 1] (last column: real row)
        losses = {}
        loss_log = []
        nums = 0
        ring = None
        layout = {}

        def check_faults():
            # (after a host synchronisation: every forward issued before it has completed) a stalled BiLSTM hand-off is an
            # error here, never NaN scores in the prediction pickle
            if hasattr(model, "check_faults"):
                model.check_faults()

        def on_half(g, n_valid):
            if not D.is_main_process():
                check_faults()
                return
            # ONE device-to-host copy of the gathered rows; everything else is host-side slicing
            host = g.view(world, self.GATHER_EVERY, layout["B"], -1)[:, :n_valid].detach().cpu().numpy()
            check_faults()
            for r in range(world):
                rec_rows[r].append(host[r].reshape(n_valid * layout["B"], -1))

        G = int(self.cfg.hip.get("batch_requests", 1)) if "hip" in self.cfg else 1
        dev = torch.device(self.device)

        def host_T(bts):
            """Longest sentence of the group from the host copies of the lengths (None: they live on the device)."""
            ls = [bt.get("srl_arg_word_mask_len") for bt in bts]
            if any(l is None or l.is_cuda for l in ls):
                return None
            return max(int(l.max()) for l in ls)

        def batches():
            """(device batch, host batches, sizes): the loader's batches, `batch_requests` of them at a time as one
            (dynamic batching: rows never interact; concatenated on the device: 35 MB per group on the host cost 10 ms)."""
            pend, hosts = [], []
            for dbt, hbt in DevicePrefetcher(dl, dev, depth=2, hold=max(1, G)):
                if dev.type != "cuda" or not all(v.is_cuda for v in dbt.values()):
                    dbt = {k: v.to(dev) for k, v in dbt.items()}
                if G <= 1:
                    yield dbt, [hbt], [next(iter(dbt.values())).shape[0]]
                    continue
                pend.append(dbt)
                hosts.append(hbt)
                if len(pend) == G:
                    yield {k: torch.cat([p_[k] for p_ in pend], dim=0) for k in pend[0]}, hosts, [next(iter(p_.values())).shape[0] for p_ in pend]
                    pend, hosts = [], []
            if pend:
                yield {k: torch.cat([p_[k] for p_ in pend], dim=0) for k in pend[0]}, hosts, [next(iter(p_.values())).shape[0] for p_ in pend]

        for batch, hosts, sizes in batches():
            b = next(iter(batch.values())).shape[0]
            with torch.no_grad():
                T = host_T(hosts) if getattr(model, "supports_T_hint", False) else None
                out = model(batch, T=T) if T is not None else model(batch)
                if loss_fn is not None:
                    # the loss of every loader batch on its own rows: the reference averages per-batch means (:113-127)
                    lo = 0
                    for sz in sizes:
                        if len(sizes) == 1:
                            ld = loss_fn(out, batch)
                        else:
                            ld = loss_fn({k: v[lo:lo + sz] for k, v in out.items() if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == b},
                                         {k: v[lo:lo + sz] for k, v in batch.items()})
                        # (kept as 0-dim device tensors and reduced ONCE behind the loop: a running `+= v.double() * sz` is three
                        # tiny launches per key and batch)
                        loss_log.append(({k: v.detach() for k, v in ld.items()}, sz))
                        nums += sz
                        lo += sz
            rec = self._records(out, batch)
            meta = [k for k in self.META_KEYS if k in batch]
            if ring is None:
                # rows of one ring entry: the loader's batch size x the requests served per forward - not whatever this rank's
                # FIRST batch happens to hold (a wrapped-around shard can start with the short tail batch), and the same on
                # every rank (the all-gather's sizes must agree)
                rows_ring = max(rec.shape[0], int(self.cfg.train.get("bsv", 0) or 0) * max(1, G))
                if D.get_world_size() > 1:
                    t = torch.tensor([rows_ring], dtype=torch.int64, device=rec.device)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    rows_ring = int(t.item())
                layout.update(B=rows_ring, rw=rec.shape[1], ncmp=batch["new_srl_idxs"].size(1),
                              nsrl=out["mdl_outs_eval"].shape[2], meta=meta,
                              meta_w={k: int(batch[k].numel() // rec.shape[0]) for k in meta},
                              meta_1d={k: batch[k].dim() == 1 for k in meta})
                ring = D.RecordRing(rows_ring, rec.shape[1], self.GATHER_EVERY, rec.device, on_half=on_half)
            nb = rec.shape[0]
            assert nb <= layout["B"], (f"batch of {nb} queries, the exchange ring holds {layout['B']} per entry "
                                       "(cfg.train.bsv x cfg.hip.batch_requests, or the first batch if larger)")
            # metadata of the entry's rows from the HOST batches (a device-resident loader batch is read back: a sync per batch)
            m = np.zeros((layout["B"], sum(layout["meta_w"].values()) + 1), dtype=np.int64)
            lo = 0
            for hbt, sz in zip(hosts, sizes):
                off = 0
                for k in meta:
                    w = layout["meta_w"][k]
                    m[lo:lo + sz, off:off + w] = hbt[k].detach().cpu().numpy().reshape(sz, w)
                    off += w
                lo += sz
            m[:nb, -1] = 1                             # real rows: a short batch (validation loaders keep the tail,
            meta_rows.append(m)                        # drop_last=is_train, utils/trn_utils.py:200-203) is padded to the ring's rows
            row = rec
            if nb < layout["B"]:
                row = torch.cat([rec, rec.new_zeros(layout["B"] - nb, rec.shape[1])], dim=0)
            ring.push(row, torch.cuda.current_stream() if rec.is_cuda else None)
        if ring is not None:
            ring.flush()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        check_faults()
        # the metadata of every rank on rank 0: one exchange for the whole loop
        meta_all = None
        if meta_rows:
            mine = np.stack(meta_rows)                 # [entries, rows_ring, W + 1]
            if world > 1:
                t = torch.from_numpy(mine).to(dev)
                cnt = torch.tensor([t.shape[0], -t.shape[0]], dtype=torch.int64, device=dev)
                torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.MAX)
                assert int(cnt[0]) == -int(cnt[1]), "every rank must run the same number of validation batches (DistributedSampler pads)"
                outl = [torch.empty_like(t) for _ in range(world)]
                torch.distributed.all_gather(outl, t)
                meta_all = [o.cpu().numpy() for o in outl] if D.is_main_process() else None
            else:
                meta_all = [mine]
        # kept as numpy columns; the reference's per-query dicts of Python lists (eval_vsrl_corr.py:247-273) are never
        # built: rank 0 writes their pickle bytes directly (fast_pickle.dumps_records, byte-identical;
        # `tolist` + `pickle.dumps` of 512 queries cost 170 ms = a 3 k queries/s ceiling for the whole validation loop)
        chunks = []                                    # (rank, batch) order: the reference's merge order
        if D.is_main_process() and meta_all is not None:
            for r in range(world):
                rows = np.concatenate(rec_rows[r], axis=0) if rec_rows[r] else np.zeros((0, layout["rw"]), np.float32)
                mr = meta_all[r].reshape(-1, meta_all[r].shape[-1])
                assert rows.shape[0] == mr.shape[0], (rows.shape, mr.shape)
                keep = mr[:, -1] > 0
                u = self.unpack(torch.from_numpy(np.ascontiguousarray(rows[keep])), layout["ncmp"], layout["nsrl"])
                cols = {"pred_boxes": u["boxes"].numpy(), "pred_scores": u["scores"].numpy(), "pred_cmp": u["indexs"].numpy()}
                off = 0
                for k in layout["meta"]:
                    w = layout["meta_w"][k]
                    mk = mr[keep][:, off:off + w]
                    cols[self.META_NAMES[k]] = np.ascontiguousarray(mk[:, 0] if layout["meta_1d"][k] else mk)
                    off += w
                chunks.append(cols)
        merged = {k: np.concatenate([c[k] for c in chunks], axis=0) for k in chunks[0]} if chunks else {}
        if loss_log:
            wts = torch.tensor([float(sz) for _, sz in loss_log], dtype=torch.float64, device=next(iter(loss_log[0][0].values())).device)
            for k in loss_log[0][0]:
                losses[k] = (torch.stack([d[k] for d, _ in loss_log]).double() * wts).sum()
        val_loss = {k: (v / max(1, nums)).float() for k, v in losses.items()}
        if D.get_world_size() > 1:
            for k in sorted(val_loss):                 # as reduce_dict in the reference (utils/trn_utils.py:61-90)
                t = val_loss[k].clone()
                torch.distributed.all_reduce(t)
                val_loss[k] = t / world
        val_acc = {k: torch.tensor(0.0) for k in self.met_keys}
        if D.is_main_process() and pred_path is not None:
            fname = Path(pred_path) / f"{dl_name}_{rank}.pkl"
            fname.parent.mkdir(parents=True, exist_ok=True)
            with open(fname, "wb") as f:
                f.write(fast_pickle.dumps_records(merged) if merged else pickle.dumps([]))
            if self.grnd_eval is not None:
                acc = self.grnd_eval.eval_ground_acc(fname)
                val_acc = {k: torch.tensor(v) for k, v in acc.items() if k in self.met_keys}
        D.synchronize()
        return val_loss, val_acc


class EvaluatorSEP(Evaluator):
    conc_type = "sep"


class EvaluatorTEMP(Evaluator):
    conc_type = "temp"


class EvaluatorSPAT(Evaluator):
    conc_type = "spat"

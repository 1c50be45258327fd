"""Grounding metrics over the evaluator's prediction pickle.

Same surface as the reference's code/eval_fn_corr.py - `GroundEval_SEP / GroundEval_TEMP / GroundEval_SPAT
(cfg, comm)`, `.eval_ground_acc(predict_file, split_type='valid')` -> the metric dictionary the
`Evaluator` classes read (`avg1` accuracy, `avg1_cons` consistency, `avg1_vidf` video accuracy,
`avg1_strict` strict accuracy, their `avg2` / `macro_*` forms and the per-verb breakdown) - and the same
input files (`cfg.ds.val_ds4_inds` csv of SRL sentences, `cfg.ds.anet_ent_annot_file` json of boxes,
`cfg.train.prob_thresh`). This is host-side bookkeeping (a few hundred IoUs per query): plain numpy, no
GPU; what it needs from the device path is the record format, which `Evaluator.forward` writes.

Definitions (reference line numbers in parentheses). A query = one SRL sentence whose video is one of
`ncmp` compared videos (`targ_cmp`); an argument counts when the annotation marks it groundable
(`req_cls_pats_mask[a][1] == 1`) and it is *correct* when

  SEP  (eval_fn_corr.py:302-346)  the video chosen for the whole query (most frequent entry of `pred_cmp`)
       is the target and, in some annotated frame of the argument, the predicted box has IoU > 0.5 with
       the annotated one and a score above `prob_thresh`;
  TEMP (:492-596)  every compared video behaves: the target video has such a frame, every other (unmasked)
       video has NO annotated frame scored above the threshold;
  SPAT (:627-747)  every frame behaves: in frames where the argument is annotated the chosen video is the
       target and the box matches one of that frame's annotated boxes (shifted by 720 px per video slot)
       above the threshold; in the other frames no foreign video is chosen above the threshold.

Per query: `res` = correct arguments, `tot` = groundable arguments, consistency / video accuracy from
the per-argument video decisions (:359-369, :476-490, :603-625), strict = all arguments correct (:279-286).
`avg1` = sum(res) / sum(tot), `avg2` = mean(res / tot), `macro_*` = mean of the per-verb averages
(:180-193, :246-251). Pinned against the reference classes on synthetic annotation sets:
oracle/make_golden_metrics.py -> tests/golden/metrics/, tests/test_metrics.py.
"""
from __future__ import annotations

import ast
import csv
import json
import pickle
from collections import OrderedDict

import numpy as np

__all__ = ["GroundEval_Corr", "GroundEval_SEP", "GroundEval_TEMP", "GroundEval_SPAT", "box_iou_f32", "main"]


def box_iou_f32(a, b) -> np.float32:
    """IoU of two x1y1x2y2 boxes in float32, operation for operation utils/box_utils.py:25-52 (no +1)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    w = np.maximum(np.minimum(a[2], b[2]) - np.maximum(a[0], b[0]), np.float32(0))
    h = np.maximum(np.minimum(a[3], b[3]) - np.maximum(a[1], b[1]), np.float32(0))
    inter = np.float32(w * h)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.float32(inter / np.float32(np.float32(area_a + area_b) - inter))


def _most_common(values):
    """First-seen value with the highest count (collections.Counter.most_common order)."""
    counts = OrderedDict()
    for v in values:
        counts[v] = counts.get(v, 0) + 1
    best, best_n = None, -1
    for v, n in counts.items():
        if n > best_n:
            best, best_n = v, n
    return best


def _read_table(path):
    """csv -> list of dict rows (row number = the sentence index); '[...]' cells become python lists."""
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    for r in rows:
        for k, v in r.items():
            if isinstance(v, str) and v[:1] == "[":
                r[k] = ast.literal_eval(v)
    return rows


def _get(cfg, *path):
    for p in path:
        cfg = cfg[p] if isinstance(cfg, dict) and not hasattr(cfg, p) else getattr(cfg, p)
    return cfg


class GroundEval_Corr:
    """Single-video accuracy (eval_fn_corr.py:42-270): an argument's annotated boxes are checked frame by
    frame; `res` counts matching boxes."""
    RES = ("res_dict",)

    def __init__(self, cfg, comm):
        self.cfg, self.comm = cfg, comm
        self.res_dicts = list(self.RES)
        try:
            self.prob_thresh = float(_get(cfg, "train", "prob_thresh"))
        except (AttributeError, KeyError):
            self.prob_thresh = 0.2                       # configs/anet_srl_cfg.yml:121
        self.prepare_gt("valid")
        self.after_init()

    def after_init(self):
        return

    # ---- annotations ------------------------------------------------------------------------------------
    def prepare_gt(self, split_type="valid"):
        if split_type not in ("valid", "test"):
            raise NotImplementedError(split_type)
        self.srl_annots1 = _read_table(_get(self.cfg, "ds", "val_ds4_inds"))
        with open(_get(self.cfg, "ds", "anet_ent_annot_file")) as f:
            self.anet_annots = json.load(f)
        want = "val" if split_type == "valid" else "test"
        self.srl_annots = [i for i, r in enumerate(self.srl_annots1) if r["vt_split"] == want]
        self._gt_cache = {}

    def gt_of(self, sent_idx):
        """-> (boxes [n,4] int64, frames [n] int64) of the segment the sentence belongs to."""
        if sent_idx not in self._gt_cache:
            vid, seg = self.srl_annots1[sent_idx]["vid_seg"].split("_segment_")
            ann = self.anet_annots[vid]["segments"][str(int(seg))]
            boxes = np.asarray(ann["bbox"], dtype=np.int64).reshape(-1, 4)
            frames = np.asarray(ann["frm_idx"], dtype=np.int64)
            assert len(boxes) == len(frames)
            self._gt_cache[sent_idx] = (boxes, frames)
        return self._gt_cache[sent_idx]

    @staticmethod
    def prepare_preds(predict_file):
        with open(predict_file, "rb") as f:
            recs = pickle.load(f)
        out = {}
        for r in recs:                                   # first record of a sentence wins
            out.setdefault(int(r["idx_sent"]), r)
        return out

    # ---- one query ----------------------------------------------------------------------------------------
    def eval_one_sent_idx(self, rec, sent_idx):
        row = self.srl_annots1[sent_idx]
        boxes, frames = self.gt_of(sent_idx)
        nargs = len(row["req_args"])
        pb = rec["pred_boxes"][:nargs]
        res = tot = 0
        for a, (_, has_box, inds) in enumerate(row["req_cls_pats_mask"]):
            if has_box != 1:
                continue
            tot += 1
            if a >= len(pb):
                continue
            for bi in inds:
                if box_iou_f32(pb[a][int(frames[bi])][:4], boxes[bi]) > 0.5:
                    res += 1
        return {"res_dict": res, "tot_dict": tot} if tot else None

    # ---- aggregation --------------------------------------------------------------------------------------
    def _averages(self, per_query):
        """per_query: {idx: {name: value, 'tot_dict': n}} -> (avg1, avg2) per result name."""
        keys = sorted(per_query)
        tot = np.array([per_query[k]["tot_dict"] for k in keys])
        avg1, avg2 = {}, {}
        for name in self.res_dicts:
            v = np.array([per_query[k][name] for k in keys])
            avg1[name] = v.sum() / tot.sum()
            avg2[name] = np.divide(v, tot).mean()
        return avg1, avg2

    def post_proc_final(self, out):
        return out

    def eval_ground_acc(self, predict_file, split_type="valid"):
        self.prepare_gt(split_type)
        preds = self.prepare_preds(predict_file)
        allq, by_verb = {}, OrderedDict()
        for sent_idx in self.srl_annots:
            rec = preds[sent_idx]                         # every validation sentence must have been predicted
            if "idx_verbs" in rec:
                assert rec["idx_verbs"][rec["targ_cmp"]] == sent_idx
            q = self.eval_one_sent_idx(rec, sent_idx)
            if q is None:
                continue
            allq[sent_idx] = q
            by_verb.setdefault(self.srl_annots1[sent_idx]["lemma_verb"], {})[sent_idx] = q
        avg1, avg2 = self._averages(allq)
        cls = {v: self._averages(q) for v, q in by_verb.items()}
        macro1 = {n: float(np.mean([c[0][n] for c in cls.values()])) for n in self.res_dicts}
        macro2 = {n: float(np.mean([c[1][n] for c in cls.values()])) for n in self.res_dicts}
        first = self.res_dicts[0]
        out = {"avg1": avg1[first], "avg2": avg2[first], "macro_avg1": macro1[first], "macro_avg2": macro2[first],
               "res_dicts_avg1": avg1, "res_dicts_macro_avg1": macro1, "classwise_dict": by_verb,
               "num_queries": len(allq)}
        return self.post_proc_final(out)


class GroundEval_SEP(GroundEval_Corr):
    RES = ("res_dict", "cons_dict", "vidf_dict", "strict_res_dict")

    def after_init(self):
        self.num_sampled_frm = int(_get(self.cfg, "ds", "num_sampled_frm"))
        self.num_prop_per_frm = _get(self.comm, "num_prop_per_frm")

    def post_proc_final(self, out):
        a, m = out["res_dicts_avg1"], out["res_dicts_macro_avg1"]
        out.update(avg1_cons=a["cons_dict"], macro_avg1_cons=m["cons_dict"], avg1_strict=a["strict_res_dict"],
                   macro_avg1_strict=m["strict_res_dict"], avg1_vidf=a["vidf_dict"], macro_avg1_vidf=m["vidf_dict"])
        return out

    # -- the three rules; each returns (correct, video decision[, score]) for one argument ------------------
    def _hit(self, box, score, gt_box):
        return bool(box_iou_f32(box[:4], gt_box) > 0.5 and score > self.prob_thresh)

    def argument(self, rec, a, targ, arg_boxes, arg_frames, frames_all, query_vid):
        if query_vid != targ:
            return False, query_vid
        pb, ps = rec["pred_boxes"][a][query_vid], rec["pred_scores"][a][query_vid]
        ok = any(self._hit(pb[int(f)], ps[int(f)], g) for g, f in zip(arg_boxes, arg_frames))
        return ok, query_vid

    def video_of(self, query_vid, per_frame):
        return query_vid

    def consistency(self, decisions, targ):
        if not decisions:
            return 0, 0
        return 1, int(decisions[0] == targ)

    def eval_one_sent_idx(self, rec, sent_idx):
        targ = int(rec["targ_cmp"])
        row = self.srl_annots1[sent_idx]
        boxes, frames = self.gt_of(sent_idx)
        frames_all = [self.gt_of(int(v))[1] for v in rec["idx_verbs"]]
        query_vid = _most_common([x for per_arg in rec["pred_cmp"] for x in per_arg])
        npred = len(rec["pred_boxes"])
        res = tot = 0
        decisions = []
        for a, (_, has_box, inds) in enumerate(row["req_cls_pats_mask"]):
            if has_box != 1:
                continue
            tot += 1
            if a >= npred:
                continue
            inds = np.asarray(inds, dtype=np.int64)
            ok, vid = self.argument(rec, a, targ, boxes[inds], frames[inds], frames_all,
                                    self.video_of(query_vid, rec["pred_cmp"][a]))
            decisions.append(vid)
            res += int(ok)
        if not tot:
            return None
        cons, vidf = self.consistency(decisions, targ)
        return {"res_dict": res, "tot_dict": tot, "cons_dict": tot * cons, "vidf_dict": tot * vidf,
                "strict_res_dict": int(res == tot) * tot}


class GroundEval_TEMP(GroundEval_SEP):
    def argument(self, rec, a, targ, arg_boxes, arg_frames, frames_all, query_vid):
        msk = rec["cmp_msk"]
        fired = []                                        # (video, score) of foreign videos above the threshold
        all_ok = True
        for v in range(len(rec["pred_boxes"][a])):
            if msk[v] != 1:
                continue
            pb, ps = rec["pred_boxes"][a][v], rec["pred_scores"][a][v]
            if v == targ:
                if not any(self._hit(pb[int(f)], ps[int(f)], g) for g, f in zip(arg_boxes, arg_frames)):
                    all_ok = False
            else:
                first = next((ps[int(f)] for f in frames_all[v] if ps[int(f)] > self.prob_thresh), None)
                if first is not None:
                    all_ok = False
                    fired.append((v, first))
        if all_ok:
            return True, targ
        if not fired:
            return False, -1
        best = max(range(len(fired)), key=lambda i: (fired[i][1], -i))      # highest score, first on ties
        return False, fired[best][0]

    def consistency(self, decisions, targ):
        if not decisions:
            return 0, 0
        mc = _most_common(decisions)
        cons = all(d == mc and d >= 0 for d in decisions)
        return int(cons), int(mc == targ and cons)


class GroundEval_SPAT(GroundEval_SEP):
    def video_of(self, query_vid, per_frame):
        return per_frame

    def argument(self, rec, a, targ, arg_boxes, arg_frames, frames_all, per_frame):
        nfrm = len(rec["pred_boxes"][a][0])
        shift = np.array([720 * targ, 0, 720 * targ, 0], dtype=np.int64)
        gt_at = {}
        for g, f in zip(arg_boxes, arg_frames):
            gt_at.setdefault(int(f), []).append(g + shift)
        all_ok = True
        free = []                                         # (frame, score) of the frames without annotation
        for f in range(nfrm):
            v = int(per_frame[f])
            assert rec["cmp_msk"][v] == 1
            sc, box = rec["pred_scores"][a][v][f], rec["pred_boxes"][a][v][f]
            if f in gt_at:
                if not (v == targ and any(self._hit(box, sc, g) for g in gt_at[f])):
                    all_ok = False
            else:
                if v != targ and sc > self.prob_thresh:
                    all_ok = False
                free.append((f, sc))
        if all_ok:
            return True, targ
        if not free:
            return False, -5
        best = max(range(len(free)), key=lambda i: (free[i][1], -i))
        return False, -free[best][0]                      # (the reference reports minus the frame number)

    def consistency(self, decisions, targ):
        if not decisions:
            return 0, 0
        mc = _most_common(decisions)
        return int(all(d == mc for d in decisions)), int(all(d == targ and d >= 0 for d in decisions))


def main(pred_file, cfg, comm=None, split_type="valid"):
    """Score `pred_file` for cfg.ds.conc_type; prints and returns the four headline metrics."""
    conc = _get(cfg, "ds", "conc_type")
    cls = {"sep": GroundEval_SEP, "svsq": GroundEval_SEP, "temp": GroundEval_TEMP, "spat": GroundEval_SPAT}[conc]
    exp = _get(cfg, "ds", "exp_setting")
    comm = comm or {"num_prop_per_frm": {"gt5": 5, "p100": 100}[exp]}
    out = cls(cfg, comm).eval_ground_acc(pred_file, split_type=split_type)
    res = {k: float(out[k]) for k in ("avg1", "avg1_cons", "avg1_vidf", "avg1_strict")}
    print(res)
    return res

"""Grounding-metric goldens from the REFERENCE `GroundEval_*` classes (TEST INFRASTRUCTURE; build container only).

The reference scores a prediction pickle against the dataset's annotation files
(code/eval_fn_corr.py:42-747). The dataset is absent, so this script writes a small synthetic annotation
set in the reference's own file formats -

    val_asrl_annots.csv   rows = SRL sentences: vt_split, ann_ind, vid_seg, lemma_verb, req_args,
                          req_cls_pats_mask = [(arg, has_box, [indices into the segment's boxes]), ...]
    anet_ent.json         {video: {"segments": {seg: {"bbox": [[x1,y1,x2,y2], ...], "frm_idx": [...]}}}}
    val_postproc.csv      (read by prepare_gt, never used by the metrics)

- plus, per concatenation type, a prediction pickle in the evaluator's record format
(code/eval_vsrl_corr.py:247-273; stored as compact arrays, `records()` rebuilds the lists) that mixes right and wrong boxes, scores on both sides of
`prob_thresh`, wrong videos, masked-out videos and inconsistent arguments, runs the reference class on
them and stores its metric dictionary. Everything lands in tests/golden/metrics/ (a few KB).

    python -m oracle.make_golden_metrics
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np

from oracle import ref_import

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "metrics")
NFRM, NCMP, NSRL = 10, 4, 5
PROB_THRESH = 0.2
KEYS = ("avg1", "avg2", "macro_avg1", "macro_avg2", "avg1_cons", "macro_avg1_cons", "avg1_strict",
        "macro_avg1_strict", "avg1_vidf", "macro_avg1_vidf")
VERBS = ["run", "throw", "hold", "cut"]
ARGS = ["ARG0", "ARG1", "ARG2", "ARGM-LOC", "V"]


def annotation_set(seed: int, n_sent: int = 64):
    """-> (rows of the SRL csv, entity json). Sentence i lives in segment i of video i // 3."""
    rng = np.random.RandomState(seed)
    rows, ent = [], {}
    for i in range(n_sent):
        vid, seg = f"v_{i // 3:05d}", i % 3
        nbox = int(rng.randint(2, 7))
        frms = sorted(rng.choice(NFRM, size=nbox, replace=True).tolist())
        boxes = []
        for _ in range(nbox):
            x1, y1 = int(rng.randint(0, 500)), int(rng.randint(0, 300))
            boxes.append([x1, y1, x1 + int(rng.randint(40, 200)), y1 + int(rng.randint(40, 150))])
        ent.setdefault(vid, {"segments": {}})["segments"][str(seg)] = {"bbox": boxes, "frm_idx": frms}
        nargs = int(rng.randint(2, NSRL + 1))
        pats, used = [], 0
        for a in range(nargs):
            has = int(rng.rand() < 0.7) if a else 1                 # at least one groundable argument
            if has and used < nbox:
                k = int(rng.randint(1, min(3, nbox - used) + 1))
                inds = list(range(used, used + k))
                used += k
            else:
                has, inds = 0, [0]
            pats.append((ARGS[a], has, inds))
        rows.append({"vt_split": "val" if i % 4 else "test", "ann_ind": i // 3, "vid_seg": f"{vid}_segment_{seg:02d}",
                     "lemma_verb": VERBS[int(rng.randint(len(VERBS)))], "req_args": str([p[0] for p in pats]),
                     "req_cls_pats_mask": str(pats)})
    return rows, ent


def predictions(rows, ent, conc: str, seed: int):
    """One record per sentence (idx_sent = row number): the target is one of NCMP compared sentences."""
    rng = np.random.RandomState(seed)
    n = len(rows)
    recs = []
    for i in range(n):
        others = rng.choice([j for j in range(n) if j != i], size=NCMP - 1, replace=False).tolist()
        targ = int(rng.randint(NCMP))
        verbs = others[:targ] + [i] + others[targ:]
        cmp_msk = [1] * NCMP
        if conc != "spat" and rng.rand() < 0.3:                    # a masked-out (padding) video, never the target
            k = int(rng.choice([c for c in range(NCMP) if c != targ]))
            cmp_msk[k] = 0
        vid, seg = rows[i]["vid_seg"].split("_segment_")
        g = ent[vid]["segments"][str(int(seg))]
        quality = rng.rand()                                       # per query: mostly right / mixed / mostly wrong
        boxes = np.zeros((NSRL, NCMP, NFRM, 7))
        scores = np.zeros((NSRL, NCMP, NFRM))
        pcmp = np.zeros((NSRL, NFRM), dtype=np.int64)
        for s in range(NSRL):
            for c in range(NCMP):
                for f in range(NFRM):
                    x1, y1 = rng.randint(0, 500), rng.randint(0, 300)
                    b = [x1, y1, x1 + rng.randint(30, 200), y1 + rng.randint(30, 150)]
                    if c == targ and f in g["frm_idx"] and rng.rand() < 0.5 + 0.5 * quality:
                        gb = g["bbox"][g["frm_idx"].index(f)]    # near one annotated box of that frame
                        b = [gb[0] + rng.randint(-8, 9), gb[1] + rng.randint(-8, 9), gb[2] + rng.randint(-8, 9),
                             gb[3] + rng.randint(-8, 9)]
                    if conc == "spat":
                        b[0] += 720 * c; b[2] += 720 * c            # SPAT boxes live in the concatenated frame
                    boxes[s, c, f, :4] = b
                    boxes[s, c, f, 4] = f
                    hi = (c == targ) == (rng.rand() < 0.6 + 0.4 * quality)
                    scores[s, c, f] = rng.uniform(0.25, 0.95) if hi else rng.uniform(0.0, 0.18)
                    if not cmp_msk[c]:
                        scores[s, c, f] = 0.0
            for f in range(NFRM):
                if conc == "spat":
                    pcmp[s, f] = targ if rng.rand() < 0.35 + 0.65 * quality else int(rng.randint(NCMP))
                elif conc == "temp":
                    pcmp[s, f] = 0
                else:
                    pcmp[s, f] = targ if rng.rand() < 1.3 * quality else int(rng.choice([c for c in range(NCMP) if cmp_msk[c]]))
        recs.append({"pred_boxes": boxes[..., :5].astype(np.int32), "pred_scores": scores.astype(np.float32),
                     "pred_cmp": pcmp, "idx_vid": rows[i]["ann_ind"], "idx_verbs": verbs, "idx_sent": i,
                     "cmp_msk": cmp_msk, "targ_cmp": targ})
    recs.append(dict(recs[3]))                                     # a duplicate, as a second validation pass leaves
    return {k: np.stack([np.asarray(r[k]) for r in recs]) for k in recs[0]}


def predictions_corr(rows, ent, seed: int):
    """Single-video records for GroundEval_Corr: pred_boxes [nsrl][nfrm][5], idx_verbs = [idx_sent]."""
    rng = np.random.RandomState(seed)
    recs = []
    for i in range(len(rows)):
        vid, seg = rows[i]["vid_seg"].split("_segment_")
        g = ent[vid]["segments"][str(int(seg))]
        boxes = np.zeros((NSRL, 1, NFRM, 7))
        for s in range(NSRL):
            for f in range(NFRM):
                x1, y1 = rng.randint(0, 500), rng.randint(0, 300)
                b = [x1, y1, x1 + rng.randint(30, 200), y1 + rng.randint(30, 150)]
                if f in g["frm_idx"] and rng.rand() < 0.6:
                    gb = g["bbox"][g["frm_idx"].index(f)]
                    b = [gb[0] + rng.randint(-8, 9), gb[1] + rng.randint(-8, 9), gb[2] + rng.randint(-8, 9),
                         gb[3] + rng.randint(-8, 9)]
                boxes[s, 0, f, :4] = b
                boxes[s, 0, f, 4] = f
        recs.append({"pred_boxes": boxes[..., :5].astype(np.int32), "pred_scores": np.ones((NSRL, 1, NFRM), np.float32),
                     "pred_cmp": np.zeros((NSRL, NFRM), np.int64), "idx_vid": rows[i]["ann_ind"], "idx_verbs": [i],
                     "idx_sent": i, "cmp_msk": [1], "targ_cmp": 0})
    return {k: np.stack([np.asarray(r[k]) for r in recs]) for k in recs[0]}


def records(arr, conc: str):
    """Arrays of the fixture -> the evaluator's python-list records (code/eval_vsrl_corr.py:247-273)."""
    n = len(arr["idx_sent"])
    out = []
    for i in range(n):
        b = np.zeros(arr["pred_boxes"][i].shape[:-1] + (7,), dtype=np.float32)
        b[..., :5] = arr["pred_boxes"][i]
        pc = arr["pred_cmp"][i]
        if conc == "corr":                                           # [nsrl][nfrm][7]: one video, no video axis
            b = b[:, 0]
        out.append({"pred_boxes": b.tolist(), "pred_scores": arr["pred_scores"][i].astype(np.float32).tolist(),
                    "pred_cmp": (pc.astype(np.float32) if conc == "temp" else pc.astype(np.int64)).tolist(),
                    "idx_vid": int(arr["idx_vid"][i]), "idx_verbs": arr["idx_verbs"][i].tolist(),
                    "idx_sent": int(arr["idx_sent"][i]), "cmp_msk": arr["cmp_msk"][i].tolist(),
                    "targ_cmp": int(arr["targ_cmp"][i]), "perm": list(range(NCMP)), "perm_inv": list(range(NCMP))})
    return out


def write_pickle(conc: str, path: str):
    arr = np.load(os.path.join(OUT, f"preds_{conc}.npz"))
    with open(path, "wb") as f:
        pickle.dump(records(arr, conc), f, protocol=4)
    return path


def write_fixture(seed: int = 11):
    import pandas as pd
    os.makedirs(OUT, exist_ok=True)
    rows, ent = annotation_set(seed)
    pd.DataFrame(rows).to_csv(os.path.join(OUT, "val_asrl_annots.csv"), index=False)
    pd.DataFrame({"dummy": [0]}).to_csv(os.path.join(OUT, "val_postproc.csv"), index=False)
    with open(os.path.join(OUT, "anet_ent.json"), "w") as f:
        json.dump(ent, f)
    for conc in ("sep", "temp", "spat"):
        np.savez_compressed(os.path.join(OUT, f"preds_{conc}.npz"),
                            **predictions(rows, ent, conc, seed + {"sep": 1, "temp": 2, "spat": 3}[conc]))
    np.savez_compressed(os.path.join(OUT, "preds_corr.npz"), **predictions_corr(rows, ent, seed + 4))
    return rows, ent


def metric_cfg():
    ds = ref_import.Munch(val_ds4_inds=os.path.join(OUT, "val_asrl_annots.csv"),
                          val_ann_file=os.path.join(OUT, "val_postproc.csv"),
                          anet_ent_annot_file=os.path.join(OUT, "anet_ent.json"), num_sampled_frm=NFRM, do_ds4=True)
    return ref_import.Munch(ds=ds, train=ref_import.Munch(prob_thresh=PROB_THRESH))


def reference_metrics(conc: str):
    ref_import.install_stubs()
    import eval_fn_corr as ref                                      # noqa: reference module
    cls = {"sep": ref.GroundEval_SEP, "temp": ref.GroundEval_TEMP, "spat": ref.GroundEval_SPAT,
           "corr": ref.GroundEval_Corr}[conc]
    ev = cls(metric_cfg(), ref_import.Munch(num_prop_per_frm=5))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = ev.eval_ground_acc(write_pickle(conc, os.path.join(td, f"preds_{conc}.pkl")))
    res = {k: float(out[k]) for k in KEYS if k in out}
    res["classes"] = {k: [float(v[0][r][i]) for r in ev.res_dicts for i in sorted(v[1])] + [float(v[1][i]) for i in sorted(v[1])]
                      for k, v in out["classwise_dict"].items()}
    return res


def main():
    write_fixture()
    gold = {conc: reference_metrics(conc) for conc in ("sep", "temp", "spat", "corr")}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    for conc, g in gold.items():
        print(conc, {k: round(g[k], 4) for k in ("avg1", "avg1_cons", "avg1_vidf", "avg1_strict") if k in g})


if __name__ == "__main__":
    main()

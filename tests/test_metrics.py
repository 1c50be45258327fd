"""Grounding metrics (vognet-pytorch_amd/eval_fn_corr.py) against the reference `GroundEval_*` classes.

tests/golden/metrics/ holds a synthetic annotation set in the reference's file formats, prediction records
for every concatenation type and the metric dictionaries the reference classes computed on them
(oracle/make_golden_metrics.py, build container). CPU only."""
import importlib
import json
import os
import time
import types

import numpy as np
import pytest

from oracle import make_golden_metrics as G
from oracle import ref_import

M = importlib.import_module("vognet-pytorch_amd.eval_fn_corr")
EXPECTED = json.load(open(os.path.join(G.OUT, "expected.json")))
CLS = {"sep": M.GroundEval_SEP, "temp": M.GroundEval_TEMP, "spat": M.GroundEval_SPAT, "corr": M.GroundEval_Corr}


def _cfg():
    ns = types.SimpleNamespace
    return ns(ds=ns(val_ds4_inds=os.path.join(G.OUT, "val_asrl_annots.csv"),
                    val_ann_file=os.path.join(G.OUT, "val_postproc.csv"),
                    anet_ent_annot_file=os.path.join(G.OUT, "anet_ent.json"), num_sampled_frm=G.NFRM, do_ds4=True,
                    conc_type="spat", exp_setting="gt5"),
              train=ns(prob_thresh=G.PROB_THRESH))


@pytest.mark.parametrize("conc", ["sep", "temp", "spat", "corr"])
def test_metrics_match_reference(conc, tmp_path):
    ev = CLS[conc](_cfg(), {"num_prop_per_frm": 5})
    out = ev.eval_ground_acc(G.write_pickle(conc, str(tmp_path / f"preds_{conc}.pkl")))
    exp = EXPECTED[conc]
    for k in G.KEYS:
        if k in exp:
            assert out[k] == pytest.approx(exp[k], rel=0, abs=1e-12), (conc, k)
    # the per-verb, per-query counts too (res / cons / vidf / strict / tot of every scored sentence)
    assert set(out["classwise_dict"]) == set(exp["classes"])
    for verb, per_q in out["classwise_dict"].items():
        idx = sorted(per_q)
        got = [float(per_q[i][r]) for r in ev.res_dicts for i in idx] + [float(per_q[i]["tot_dict"]) for i in idx]
        assert got == exp["classes"][verb], (conc, verb)
    assert 0 < out["avg1"] < 1                                    # the fixture exercises both outcomes


def test_evaluator_builds_the_metric_class_when_annotations_exist(tmp_path):
    """Evaluator*.after_init attaches GroundEval_* exactly when cfg.ds names existing annotation files
    (reference: eval_vsrl_corr.py:154-158, 277-283, 349-351), and scores a pickle through it."""
    E = importlib.import_module("vognet-pytorch_amd.eval_vsrl_corr")
    cfg = _cfg()
    ev = E.EvaluatorSPAT(cfg, {"num_prop_per_frm": 5}, "cpu")
    assert isinstance(ev.grnd_eval, M.GroundEval_SPAT) and not isinstance(ev.grnd_eval, M.GroundEval_TEMP)
    assert isinstance(E.EvaluatorTEMP(cfg, {"num_prop_per_frm": 5}, "cpu").grnd_eval, M.GroundEval_TEMP)
    acc = ev.grnd_eval.eval_ground_acc(G.write_pickle("spat", str(tmp_path / "p.pkl")))
    assert {k: acc[k] for k in ev.met_keys} == pytest.approx({k: EXPECTED["spat"][k] for k in ev.met_keys})
    cfg.ds.val_ds4_inds = str(tmp_path / "missing.csv")
    assert E.EvaluatorSPAT(cfg, {"num_prop_per_frm": 5}, "cpu").grnd_eval is None


def test_metric_edge_rules():
    """Rules that are easy to get wrong, on hand-made records (values worked out from the definitions)."""
    ev = M.GroundEval_SPAT(_cfg(), {"num_prop_per_frm": 5})
    assert M._most_common([2, 1, 1, 2, 3]) == 2                   # ties: first seen
    assert M.box_iou_f32([0, 0, 10, 10], [5, 0, 15, 10]) == np.float32(50.0 / 150.0)
    assert not (M.box_iou_f32([0, 0, 0, 0], [0, 0, 0, 0]) > 0.5)  # 0/0: not a match
    assert ev.consistency([], 0) == (0, 0)
    assert ev.consistency([0, 0], 0) == (1, 1)
    assert ev.consistency([-3, -3], 1) == (1, 0)                  # consistent, wrong
    t = M.GroundEval_TEMP(_cfg(), {"num_prop_per_frm": 5})
    assert t.consistency([-1, -1], 0) == (0, 0)                   # "no video" is never consistent
    assert t.consistency([2, 2, 1], 2) == (0, 0)


def test_metrics_speed_vs_reference(tmp_path):
    """Same file, same result, less time than the pandas / torch-scalar loops of the reference (reported in
    DESIGN.md; only checked loosely here)."""
    path = G.write_pickle("spat", str(tmp_path / "p.pkl"))
    ev = M.GroundEval_SPAT(_cfg(), {"num_prop_per_frm": 5})
    t0 = time.perf_counter(); ev.eval_ground_acc(path); mine = time.perf_counter() - t0
    if not ref_import.available():
        pytest.skip("reference tree absent")
    ref_import.install_stubs()
    import eval_fn_corr as ref
    rv = ref.GroundEval_SPAT(G.metric_cfg(), ref_import.Munch(num_prop_per_frm=5))
    t0 = time.perf_counter(); rv.eval_ground_acc(path); theirs = time.perf_counter() - t0
    print(f"metrics: {mine*1e3:.1f} ms here, {theirs*1e3:.1f} ms reference")
    assert mine < theirs * 2

"""CPU: the oracle's restatement of LossB_TEMP / LossB_SPAT / LossB_SEP (oracle/vog_oracle.py
loss_forward) against the goldens generated from the REFERENCE loss classes (oracle/make_golden_loss.py:
tests/golden/loss__*.npz) on the reference's own forward outputs; and, in the build container, against
the reference classes run live."""
import importlib

import numpy as np
import pytest
import torch

from oracle import cases, make_golden_loss as mgl, ref_import
from oracle import vog_oracle as vo


@pytest.mark.parametrize("name", mgl.LOSS_CASES)
def test_oracle_loss_vs_reference_golden(name):
    cfg, batch, c, tg = mgl.targets_for(name)
    gl = np.load(mgl.loss_path(name))
    assert str(gl["sha_targets"]) == cases.digest(tg), "target generator drifted"
    g = np.load(cases.golden_path(name))
    out = {k: torch.from_numpy(g[k]) for k in ("mdl_outs", "vidf_outs") if k in g.files}
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    res = vo.loss_forward(oc, out, vo.to_torch({**batch, **tg}), float(cfg.loss.loss_lambda))
    assert set(res) == set(gl.files) - {"sha_targets"}
    for k, v in res.items():
        assert abs(float(v) - float(gl[k])) <= 2e-6 * abs(float(gl[k])), (k, float(v), float(gl[k]))


def test_targets_are_not_trivial():
    """The synthetic ground truth must produce positive AND negative targets, masked and unmasked
    argument slots - otherwise the loss goldens would not pin the target selection."""
    cfg, batch, c, tg = mgl.targets_for("full/cfg2_vog_spat_gt5_bs4")
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    inp = vo.to_torch({**batch, **tg})
    zero = {"mdl_outs": torch.zeros(4, 1, 5, 200)}
    base = float(vo.loss_forward(oc, zero, inp)["loss"])
    assert abs(base - 200 * np.log(2.0)) < 1e-3                   # BCE at logit 0 is target independent
    ones = {"mdl_outs": torch.ones(4, 1, 5, 200)}
    l1 = float(vo.loss_forward(oc, ones, inp)["loss"])
    # all-negative targets would give 200 * softplus(1) = 262.65; positives pull it down
    assert 200 * 0.3133 < l1 < 262.6
    assert 0 < tg["srl_arg_boxes_mask"].sum() < tg["srl_arg_boxes_mask"].size


@pytest.mark.skipif(not ref_import.available(), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("name", ["small/vog_spat", "small/vog_temp", "small/vog_sep"])
def test_oracle_loss_vs_reference_live(name):
    cfg, batch, c, tg = mgl.targets_for(name)
    torch.manual_seed(0)
    g = np.load(cases.golden_path(name))
    out = {k: torch.from_numpy(g[k]) + 0.3 * torch.randn(g[k].shape) for k in ("mdl_outs", "vidf_outs") if k in g.files}
    ref = mgl.reference_loss(cfg, c, out, {**batch, **tg})
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    res = vo.loss_forward(oc, out, vo.to_torch({**batch, **tg}), float(cfg.loss.loss_lambda))
    for k in ref:
        assert abs(float(res[k]) - float(ref[k])) <= 2e-6 * abs(float(ref[k])), k

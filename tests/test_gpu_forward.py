"""-m gpu: the HIP forward (libvog_hip through the C ABI) against the committed
reference goldens and the CPU oracle.

Tolerances (north_star: pred_boxes / pred_scores within 1e-3 rel of the
reference fp32 CPU forward):
  * pred_scores / mdl_outs_eval: max relative error <= 1e-3 where the reference
    value is non-zero, exact zero where the reference is masked to zero.
  * mdl_outs (logits): abs error <= 6e-3 (sigmoid' <= 1/4 maps that to <= 1.5e-3
    abs on scores ~0.5; the binding bound is the relative one above).
  * pred_boxes: an arg-max gather. Equal to the reference except where the
    reference's top-2 proposals of that frame are within 2e-3 of each other
    (a flip there is inside the score tolerance, SURVEY.md hard-part 2).
  * indexs: same rule (arg-max over videos).
Budget (round 5): `cfg.hip.tx_dtype = auto` = f16 operands everywhere (oracle-simulated at 1.8e-4 on cfg 2, 5.7e-4 at the edge
of its envelope, wq / wk x 12), the fp32 path beyond (`precise.py`); bf16 transformers (what BASELINE.json's config 2 names and
bench.py times) are selectable and hold the bound only for near-uniform attention: 4.5e-4 simulated at random-init scale, 2e-3 at
wq / wk x 8 (tests/test_quant_budget.py, DESIGN.md section 2 "envelope").
"""
import ctypes
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import cases
from tests.gpu_util import L, build_engine, engine_mod, oracle_run, rel_err

pytestmark = pytest.mark.gpu

FULL = [n for n in cases.CASES if n.startswith("full/") and "p100" not in n]
SMALL = [n for n in cases.CASES if n.startswith("small/")]
SHARP = [n for n in cases.CASES if cases.CASES[n].get("sharp")]
# random-init scale, single-layer stacks: the cases bf16 transformers are specified for (3-layer stacks: 1.0-1.1e-3 measured)
BF16_OK = [n for n in FULL if n not in SHARP and "3layers" not in n]
# round 6: the cases `auto` runs with hi + lo operands / the fp32 path; the launch-structure variants below (step-launch BiLSTM,
# unfused tail, pairing) keep to the others (same kernels, one plan: the suite's time goes into new coverage instead)
R6 = ["full/cfg2_sharp24", "full/cfg2_sharp32", "full/cfg2_sharp48", "full/cfg3_sharp16", "full/cfg5_sharp16",
      "full/vgrnd_spat_sharp16", "full/vog_sep_sharp16"]
FULL_VARIANTS = [n for n in FULL if n not in R6 and n not in ("full/cfg2_sharp12", "full/cfg2_sharp16", "full/vog_spat_3layers_sharp8")]
SPLIT_CASES = ["full/cfg2_sharp12", "full/cfg2_sharp16", "full/cfg2_sharp24", "full/cfg2_sharp32", "full/cfg3_sharp16",
               "full/cfg5_sharp16", "full/vog_sep_sharp16", "full/vog_spat_3layers_sharp8"]
# (full/vgrnd_spat_sharp16: VidGrnd has obj_tx only - sharpness 12.3, inside the f16 envelope; it runs with the FULL list)


def _check_against(name, out, pred, g, ora, tol_rel, tol_logit):
    msg = []
    lo = out["mdl_outs"].cpu().numpy()
    ev = out["mdl_outs_eval"].cpu().numpy()
    assert lo.shape == g["mdl_outs"].shape
    e_logit = float(np.abs(lo - g["mdl_outs"]).max())
    nz = g["mdl_outs_eval"] != 0
    assert np.all(ev[~nz] == 0), "masked entries must be exactly zero"
    e_eval = float(rel_err(ev[nz], g["mdl_outs_eval"][nz]).max()) if nz.any() else 0.0
    sc = pred["scores"].cpu().numpy()
    snz = g["scores"] != 0
    e_sc = float(rel_err(sc[snz], g["scores"][snz]).max()) if snz.any() else 0.0
    msg.append(f"{name}: logit abs {e_logit:.2e} eval rel {e_eval:.2e} scores rel {e_sc:.2e}")
    for k in ("vidf_outs", "fin_scores", "fin_scores_loss"):
        if k in g.files:
            e = float(np.abs(out[k].cpu().numpy() - g[k]).max())
            msg.append(f"  {k} abs {e:.2e}")
            assert e <= 4e-3, (k, e)
    print("\n".join(msg))
    assert e_logit <= tol_logit, msg
    assert e_eval <= tol_rel and e_sc <= tol_rel, msg
    # boxes: equal, or a near-tie flip
    bx = pred["boxes"].cpu().numpy()
    diff = np.any(bx != g["boxes"], axis=-1)
    nflip = int(diff.sum())
    print(f"  box flips {nflip} of {diff.size}")
    # (round 6) the NUMBER of flips is bounded too: 0.5 % of the boxes of a case (and at least the 2 a handful-of-boxes case may
    # legitimately have); north_star asks pred_boxes within 1e-3 - a flip is only ever a near tie, but not any number of them
    assert nflip <= max(2, int(0.005 * diff.size)), f"{name}: {nflip} of {diff.size} boxes differ from the reference"
    if nflip:
        # the score of the box we picked must be within tolerance of the reference max
        bad = rel_err(sc[diff], g["scores"][diff]) > 2e-3
        assert not bad.any(), f"{name}: {int(bad.sum())} box flips outside the score tolerance"
    idx = pred["indexs"].cpu().numpy()
    assert idx.dtype == g["indexs"].dtype and idx.shape == g["indexs"].shape
    d_idx = idx != g["indexs"]
    if d_idx.any():
        # pred_cmp is an arg-max over the ncmp videos (spat: of the per-frame max score,
        # eval_vsrl_corr.py:399-411; sep: of fin_scores, :172,213-214): a flip is only allowed where
        # the REFERENCE score of the video we picked is within 2e-3 (relative) of the reference maximum
        if "fin_scores" in g.files:
            ref = np.broadcast_to(g["fin_scores"][:, None, None, :], idx.shape + (g["fin_scores"].shape[1],))
        else:
            ref = np.moveaxis(g["scores"], 2, -1)                      # [B, nsrl, nfrm, ncmp]
        picked = np.take_along_axis(ref, idx[..., None].astype(np.int64), -1)[..., 0]
        best = ref.max(-1)
        bad = d_idx & (rel_err(picked, best) > 2e-3)
        assert not bad.any(), f"{name}: {int(bad.sum())} pred_cmp flips outside the near-tie tolerance"
    return nflip


def _run(name, tx_dtype=None, graph=False, cached=True):
    eng, cfg, sd, batch, c, dev = build_engine(name, tx_dtype, cached=cached)
    before = {k: v.clone() for k, v in dev.items()}
    if graph:
        slot = eng.make_slot(dev, graph=True)
        slot.launch()
        torch.cuda.synchronize()
        out = slot.launch()
    else:
        out = eng.forward(dev)
    torch.cuda.synchronize()
    for k in before:                       # inputs are borrowed, never modified
        assert torch.equal(before[k], dev[k]), k
    ncmp = batch["new_srl_idxs"].shape[1]
    pred = eng.unpack_pred(out["pred_rec"], ncmp)
    g = np.load(cases.golden_path(name))
    return out, pred, g, (cfg, sd, batch, c)


@pytest.mark.parametrize("name", FULL)
def test_forward_full_vs_reference_golden(name):
    out, pred, g, _ = _run(name)
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


@pytest.mark.parametrize("name", SMALL)
def test_forward_small_vs_reference_golden(name):
    out, pred, g, _ = _run(name)
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


@pytest.mark.parametrize("name", BF16_OK)
def test_forward_full_bf16_vs_reference_golden(name):
    """bf16 transformers - BASELINE.json's config 2 as written, the operand type bench.py times - inside their envelope."""
    out, pred, g, _ = _run(name, tx_dtype="bf16")
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


def test_precision_plan_follows_attention_sharpness():
    """`auto` (round 5): f16 kernels inside their envelope, the fp32 path beyond it, decided per checkpoint in load_state_dict
    from the weights alone (engine.attention_sharpness); explicit requests are honoured, with a warning outside the envelope."""
    import warnings
    E = engine_mod
    seen = {}
    for name in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg2_sharp8", "full/cfg2_sharp10", "full/cfg2_sharp12", "full/cfg2_sharp16",
                 "full/cfg2_sharp32", "full/cfg2_sharp48"):
        eng, *_ = build_engine(name, cached=True)      # (the golden tests' engines: same key)
        seen[name] = (eng.sharpness, eng.plan)
        assert eng.desc.tx_dtype == L.VOG_F16 and (eng.precise is not None) == (eng.plan == "f32")
    print(seen)
    assert seen["full/cfg2_vog_spat_gt5_bs4"][0] < 1 and seen["full/cfg2_vog_spat_gt5_bs4"][1] == "f16"
    assert E.BF16_SHARPNESS_MAX < seen["full/cfg2_sharp8"][0] < E.F16_SHARPNESS_MAX and seen["full/cfg2_sharp8"][1] == "f16"
    assert seen["full/cfg2_sharp10"][0] < E.F16_SHARPNESS_MAX and seen["full/cfg2_sharp10"][1] == "f16"
    # round 6: past the f16 envelope `auto` runs hi + lo f16 operands (three MFMAs for what feeds the logits), the fp32 path only
    # past THAT plan's envelope
    assert E.F16_SHARPNESS_MAX < seen["full/cfg2_sharp12"][0] and seen["full/cfg2_sharp12"][1] == "split"
    assert seen["full/cfg2_sharp16"][1] == "split"
    assert seen["full/cfg2_sharp32"][0] < E.SPLIT_SHARPNESS_MAX and seen["full/cfg2_sharp32"][1] == "split"
    assert seen["full/cfg2_sharp48"][0] > E.SPLIT_SHARPNESS_MAX and seen["full/cfg2_sharp48"][1] == "f32"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng, *_ = build_engine("full/cfg2_sharp8", "bf16")
        assert eng.precise is None and eng.desc.tx_dtype == L.VOG_BF16
        assert any("envelope" in str(x.message) for x in w)
    eng, *_ = build_engine("full/cfg2_vog_spat_gt5_bs4", "f32")
    assert eng.precise is not None


@pytest.mark.parametrize("name", ["full/cfg2_vog_spat_gt5_bs4", "full/cfg5_vog_svsq_gt5_bs16", "full/cfg3_vog_temp_gt5_bs8",
                                  "full/cfg1_igrnd_spat_gt5_bs2", "small/vgrnd_sep", "small/vog_sep_cmpmsk", "small/vog_spat_3layers"])
def test_forward_fp32_path_vs_reference_golden(name):
    """tx_dtype = f32: the precise path (fp32 kernels of csrc/backward.hip + the exact heads) through the same engine surface,
    eager and from a graph slot - an order of magnitude inside the bound."""
    out, pred, g, _ = _run(name, tx_dtype="f32")
    _check_against(name, out, pred, g, None, tol_rel=5e-5, tol_logit=1e-4)
    out, pred, g, _ = _run(name, tx_dtype="f32", graph=True)
    _check_against(name, out, pred, g, None, tol_rel=5e-5, tol_logit=1e-4)


def test_precision_plan_for_deep_stacks():
    """3-layer stacks: the f16 envelope ends at sharpness 5 (errors compound over sharp layers): x 4 stays on the f16 kernels,
    x 8 - inside the single-layer envelope - runs the fp32 path; both hold the reference golden (run by the FULL list too)."""
    E = engine_mod
    eng, *_ = build_engine("full/vog_spat_3layers_sharp4", cached=True)
    assert eng.plan == "f16" and eng.sharpness < E.F16_SHARPNESS_MAX_DEEP
    eng, *_ = build_engine("full/vog_spat_3layers_sharp8", cached=True)      # round 6: hi + lo operands (fp32 path in round 5)
    assert eng.plan == "split" and E.F16_SHARPNESS_MAX_DEEP < eng.sharpness < E.SPLIT_SHARPNESS_MAX_DEEP
    eng, *_ = build_engine("full/cfg2_sharp8", cached=True)
    assert eng.plan == "f16"


@pytest.mark.parametrize("name", SPLIT_CASES)
def test_forward_hi_lo_plan_vs_reference_golden(name):
    """Round 6: checkpoints past the f16 envelope (wq / wk x 12 ... x 32; 3-layer stacks x 8; temp / svsq / sep / VidGrnd x 16) on
    the hi + lo plan `auto` picks for them - the fast kernels with three MFMAs for everything that feeds attention logits - against
    the reference goldens, eager and from a graph slot. (Round 5 ran these on the fp32 path, 34 x slower.)"""
    eng, *_ = build_engine(name, cached=True)
    assert eng.plan == "split", (eng.plan, eng.sharpness)
    out, pred, g, _ = _run(name)
    nf = _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)
    out2, pred2, g, _ = _run(name, graph=True)
    _check_against(name, out2, pred2, g, None, tol_rel=1e-3, tol_logit=6e-3)
    assert torch.equal(out["mdl_outs"], out2["mdl_outs"])


@pytest.mark.parametrize("name", ["full/cfg2_sharp8", "full/cfg3_vog_temp_gt5_bs8",
                                  "full/cfg5_vog_svsq_gt5_bs16", "full/vog_sep_gt5_bs4_ragged", "full/vgrnd_spat_sharp16",
                                  "full/vog_spat_gt5_bs4_3layers"])
def test_forward_hi_lo_forced_vs_reference_golden(name):
    """tx_dtype = split on checkpoints that do not need it (every model kind / conc type at full size): same goldens, errors at or
    below the f16 plan's. Small-dim models have no hi + lo kernels (vog_ctx_split_supported) and say so."""
    eng, *_ = build_engine(name, "split")
    assert eng.plan == "split"
    out, pred, g, _ = _run(name, tx_dtype="split")
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)
    if name == "full/cfg2_vog_spat_gt5_bs4":
        with pytest.raises(L.VogError, match="hi \\+ lo kernels"):
            build_engine("small/vog_spat", "split")


def test_logit_scale_guard_raises_the_plan():
    """The run-time side of the precision plan: the attention kernels report the largest |logit| they saw (vog_batch.stats, folded
    into pinned host memory by the prediction head). A checkpoint whose weight statistic UNDER-reports its logit scale (here:
    the statistic is patched to 0, so `auto` plans plain f16 for wq / wk x 16) is caught on its first batch: a warning, the plan
    raised to hi + lo operands, and the next forward is inside the bound."""
    import warnings
    E = engine_mod
    name = "full/cfg2_sharp16"
    real = E.attention_sharpness
    E.attention_sharpness = lambda *a, **k: 0.0
    try:
        eng, cfg, sd, batch, c, dev = build_engine(name)
    finally:
        E.attention_sharpness = real
    assert eng.plan == "f16"
    g = np.load(cases.golden_path(name))
    nz = g["mdl_outs_eval"] != 0
    out = eng.forward(dev)
    torch.cuda.synchronize()
    e0 = float(rel_err(out["mdl_outs_eval"].cpu().numpy()[nz], g["mdl_outs_eval"][nz]).max())
    lo, lm = eng.observed_logit_max()
    print(f"{name} planned f16: eval rel {e0:.2e}; observed |logit| max obj {lo:.1f} mul {lm:.1f} nats")
    assert max(lo, lm) > E.F16_LOGIT_MAX
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert eng.check_logit_scale() is False
        assert any("nats observed" in str(x.message) for x in w)
    assert eng.plan == "split"
    out = eng.forward(dev)
    torch.cuda.synchronize()
    e1 = float(rel_err(out["mdl_outs_eval"].cpu().numpy()[nz], g["mdl_outs_eval"][nz]).max())
    print(f"   re-planned {eng.plan}: eval rel {e1:.2e}")
    assert e1 < 1e-3 < e0
    assert eng.check_logit_scale() is True


def test_observed_logit_scale_of_the_goldens():
    """What the kernels report on the sharpened goldens (the numbers behind engine.F16_LOGIT_MAX / SPLIT_LOGIT_MAX): every case
    inside a plan's envelope stays under that plan's logit limit, so `check_logit_scale` never fires on them."""
    E = engine_mod
    rows = []
    for name in ["full/cfg2_vog_spat_gt5_bs4", "full/cfg2_sharp8", "full/cfg2_sharp10", "full/cfg2_sharp12", "full/cfg2_sharp16",
                 "full/cfg2_sharp24", "full/cfg2_sharp32", "full/cfg3_sharp8", "full/cfg5_sharp8", "full/vog_spat_3layers_sharp8",
                 "full/cfg4_p100_sharp8"]:      # (p100: the long-sequence kernels report row-reference magnitudes, a lower bound)
        eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
        eng.forward(dev)
        torch.cuda.synchronize()
        lo, lm = eng.observed_logit_max()
        rows.append((name, eng.plan, eng.sharpness, lo, lm))
        print(f"{name:36s} plan {eng.plan:5s} sharpness {eng.sharpness:7.1f}  |logit| max obj {lo:8.1f} mul {lm:8.1f}")
        assert eng.check_logit_scale(escalate=False), (name, eng.plan, lo, lm)
        assert lm > 0 and (lo > 0 or cfg.mdl.name == "igrnd")


def test_f16_just_outside_its_envelope_still_inside_the_bound():
    """wq / wk x 12 (sharpness 28.5; `auto` already runs fp32 there): f16 forced. Measured 9.0e-4 - the envelope's margin."""
    name = "full/cfg2_sharp12"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, pred, g, _ = _run(name, tx_dtype="f16")
    _check_against(name, out, pred, g, None, tol_rel=1.2e-3, tol_logit=6e-3)


def test_bf16_leaves_the_bound_where_f16_holds_it():
    """The envelope is real: at wq / wk x 8 (attention logit std ~1.7) bf16 transformers miss 1e-3 on the reference golden
    while the default plan holds it (the reason `auto` no longer picks bf16)."""
    name = "full/cfg2_sharp8"
    out, pred, g, _ = _run(name, tx_dtype="bf16")
    nz = g["mdl_outs_eval"] != 0
    e_bf = float(rel_err(out["mdl_outs_eval"].cpu().numpy()[nz], g["mdl_outs_eval"][nz]).max())
    out, pred, g, _ = _run(name)
    e_h = float(rel_err(out["mdl_outs_eval"].cpu().numpy()[nz], g["mdl_outs_eval"][nz]).max())
    print(f"{name}: bf16 {e_bf:.2e}  auto {e_h:.2e}")
    assert e_h < 1e-3 < e_bf


@pytest.mark.parametrize("name", ["full/cfg2_ragged", "small/vog_spat_r128", "small/vog_sep_r64"])
def test_forward_persistent_lstm_layer(name):
    """The opt-in one-launch-per-layer LSTM (cross-workgroup hand-off through agent-scope
    atomics; 64 / 4 / 2 workgroups per direction here) must agree with the T-launch path."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("lstm_persistent", 0)
    a = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in eng.forward(dev).items()}
    eng.set_option("lstm_persistent", 1)
    b = eng.forward(dev)
    torch.cuda.synchronize()
    # same arithmetic, different fp32 summation order of the recurrent dot products and (fused_ih, on
    # by default with the persistent kernel) of the input projections; logits are O(1..10), the golden
    # tests allow 6e-3 on them
    dl = (a["mdl_outs"] - b["mdl_outs"]).abs().max().item()
    assert dl < 1.5e-3, dl
    ncmp = batch["new_srl_idxs"].shape[1]
    pa, pb = eng.unpack_pred(a["pred_rec"], ncmp), eng.unpack_pred(b["pred_rec"], ncmp)
    ds = (pa["scores"] - pb["scores"]).abs().max().item()
    assert ds < 4e-4, ds


@pytest.mark.parametrize("name", FULL_VARIANTS[::2] + ["small/vog_spat", "small/vgrnd_sep", "small/edge_temp_len1",
                                              "small/edge_sep_maxlen"])
def test_forward_step_launch_lstm_vs_reference_golden(name):
    """The step-launch BiLSTM (lstm_persistent = 0: lowest latency fallback, any number in flight)
    against the same reference goldens as the default persistent layer kernel."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("lstm_persistent", 0)
    out = eng.forward(dev)
    torch.cuda.synchronize()
    ncmp = batch["new_srl_idxs"].shape[1]
    pred = eng.unpack_pred(out["pred_rec"], ncmp)
    g = np.load(cases.golden_path(name))
    tol = (1e-3, 6e-3)
    _check_against(name, out, pred, g, None, tol_rel=tol[0], tol_logit=tol[1])


@pytest.mark.parametrize("name", FULL_VARIANTS + ["small/vog_spat"])
def test_forward_unfused_tail_vs_reference_golden(name):
    """fused_tail = 0: the separate Wo / LayerNorm / FFN / lin2 / score launches (the path of every
    shape the fused kernel does not cover) against the same goldens."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("fused_tail", 0)
    out = eng.forward(dev)
    torch.cuda.synchronize()
    pred = eng.unpack_pred(out["pred_rec"], batch["new_srl_idxs"].shape[1])
    g = np.load(cases.golden_path(name))
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


@pytest.mark.parametrize("name", FULL_VARIANTS + ["full/cfg2_sharp16", "full/cfg4_vog_spat_p100_bs4"])
def test_paired_launches_equal_separate_launches(name):
    """pair_launches: two independent steps in one grid (csrc/pair.hip) run the same kernel bodies as
    the stand-alone launches -> bit-identical outputs."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("enc_lean", 1)            # (the encoder form otherwise follows the pairing decision)
    eng.set_option("pair_launches", 0)
    a = {k: v.clone() for k, v in eng.forward(dev).items() if isinstance(v, torch.Tensor)}
    eng.set_option("pair_launches", 1)
    b = eng.forward(dev)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), (name, k)


def test_cfg2_forward_is_eleven_launches():
    """The headline forward's launch structure: 11 launches, every fusion in place. A support probe that
    silently turns one of them off (round 6: the score head fell out of the mul_tx tail after `vog_tx_tail_supported` was
    tightened - two more launches, all goldens still green) must fail here, not in a bench line."""
    eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16", cached=True)
    slot = eng.make_slot(dev, graph=False)
    for k in ("prep", "lstm_layer+vis_enc", "obj_qkv", "obj_attn", "lstm_layer+obj_tail", "lstm_outproj+mul_pv", "argvec", "mul_pl",
              "mul_attn", "mul_tail", "pred_head"):
        assert eng.time_kernel(slot, k, 1) > 0, k
    for k in ("lin2", "score", "lstm_ih0", "lstm_ih1", "mul_wo", "obj_wo", "mul_ffn1", "obj_ffn1", "prop_enc", "seg_enc", "vislang",
              "mul_qkv", "lstm_step"):
        with pytest.raises(L.VogError):
            eng.time_kernel(slot, k, 1)


@pytest.mark.parametrize("name", ["full/cfg2_vog_spat_gt5_bs4", "full/cfg2_ragged", "full/cfg5_vog_svsq_gt5_bs16",
                                  "small/vog_temp"])
def test_gate_table_matches_input_projection(name):
    """round 6: layer 0's gate inputs read from the checkpoint's gate table (emb . W_ih^T + b for every token, built once at
    vog_ctx_finalize; `fused_ih` = 5: wherever it exists, the default uses it beyond 80 columns) against the projection computed per
    batch (4: in the layer kernel's prologue / a GEMM launch) - same 16-bit operands, another fp32 summation order - and against
    the reference's golden."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("fused_ih", 4)
    a = {k: v.clone() for k, v in eng.forward(dev).items() if isinstance(v, torch.Tensor)}
    eng.set_option("fused_ih", 5)
    try:
        out = eng.forward(dev)
        torch.cuda.synchronize()
        b = {k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor)}
        pred = eng.unpack_pred(out["pred_rec"], batch["new_srl_idxs"].shape[1])
        _check_against(name, out, pred, np.load(cases.golden_path(name)), None, tol_rel=1e-3, tol_logit=6e-3)
    finally:
        eng.set_option("fused_ih", 1)
    assert torch.isfinite(b["mdl_outs"]).all()
    dl = (a["mdl_outs"] - b["mdl_outs"]).abs().max().item()
    assert dl < 1.5e-3, dl                 # (often exactly 0: the gate inputs differ in the last fp32 bits, h is rounded to 16 bits)


@pytest.mark.parametrize("name", ["full/cfg2_vog_spat_gt5_bs4", "full/cfg2_ragged", "full/vog_sep_gt5_bs4_ragged",
                                  "full/cfg1_igrnd_spat_gt5_bs2"])
def test_fused_lstm_input_projection_matches_separate_gemm(name):
    """fused_ih: x W_ih^T + b computed in the persistent layer kernel's prologue vs the separate GEMM
    launch (same 16-bit operands, another fp32 summation order)."""
    eng, cfg, sd, batch, c, dev = build_engine(name, cached=True)
    eng.set_option("fused_ih", 0)
    a = {k: v.clone() for k, v in eng.forward(dev).items() if isinstance(v, torch.Tensor)}
    eng.set_option("fused_ih", 1)
    b = eng.forward(dev)
    torch.cuda.synchronize()
    assert torch.isfinite(b["mdl_outs"]).all()
    dl = (a["mdl_outs"] - b["mdl_outs"]).abs().max().item()
    assert dl < 1.5e-3, dl
    ncmp = batch["new_srl_idxs"].shape[1]
    pa, pb = eng.unpack_pred(a["pred_rec"], ncmp), eng.unpack_pred(b["pred_rec"], ncmp)
    ds = (pa["scores"] - pb["scores"]).abs().max().item()
    assert ds < 4e-4, ds


def test_forward_f16_transformers():
    """cfg 5 flavour: fp16 MFMA path with fp32 accumulate."""
    name = "full/cfg5_vog_svsq_gt5_bs16"
    out, pred, g, _ = _run(name, tx_dtype="f16")
    _check_against(name, out, pred, g, None, tol_rel=5e-4, tol_logit=3e-3)


def test_forward_graph_replay_matches_eager():
    name = "full/cfg2_vog_spat_gt5_bs4"
    out_e, pred_e, g, _ = _run(name)
    out_g, pred_g, _, _ = _run(name, graph=True)
    assert torch.equal(out_e["mdl_outs"], out_g["mdl_outs"])
    assert torch.equal(out_e["pred_rec"], out_g["pred_rec"])


def test_forward_p100_sharp_vs_reference_golden():
    """p100 with sharpened attention and heavy-tailed features (wq / wk x 8: obj logit std ~2, mul ~7 nats): the f16 fixed-reference
    attention (attn_tile2) and the guarded E x F attention - whichever of their fallbacks the data sends them to - against the
    reference golden."""
    name = "full/cfg4_p100_sharp8"
    out, pred, g, _ = _run(name)
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


def test_forward_p100_vs_reference_golden():
    name = "full/cfg4_vog_spat_p100_bs4"
    out, pred, g, _ = _run(name)
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


def test_forward_p100_bf16_vs_reference_golden():
    """cfg 4 in the operand type bench.py times it in (`--workload cfg4`: tx = bf16, 7.6e-4 in bench_r5_cfg4.json's parity block):
    the p100 golden with bf16 transformers."""
    name = "full/cfg4_vog_spat_p100_bs4"
    out, pred, g, _ = _run(name, tx_dtype="bf16")
    _check_against(name, out, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)


@pytest.mark.parametrize("fused", [1, 0])
def test_stages_vs_oracle(fused):
    """Stage-by-stage localisation on cfg 2 (which kernel is off, if any), with the fused encoder
    tails (default) and with the unfused GEMM / LayerNorm launches."""
    name = "full/cfg2_vog_spat_gt5_bs4"
    eng, cfg, sd, batch, c, dev = build_engine(name)
    eng.set_option("fused_tail", fused)
    out = eng.forward(dev)
    torch.cuda.synchronize()
    ora = oracle_run(cfg, sd, batch, c, keep_stages=True)
    st = ora["stages"]
    B, ncmp, T = 4, 4, int(batch["srl_arg_word_mask_len"].max())
    tok = eng.stage(B, ncmp, T, "tok", torch.int32, (B, T)).cpu()
    assert torch.equal(tok.long(), st["tokens"][:, :T])
    full = eng.stage(B, ncmp, T, "full", torch.float32, (B * T + 16, 256)).cpu()
    e = (full[: B * T].view(B, T, 256) - st["lstm_full_output"]).abs().max().item()
    print("lstm_full_output abs err", e)
    assert e < 5e-3
    e = (full[B * T: B * T + B] - st["final_hidden"]).abs().max().item()
    print("final_hidden abs err", e)
    assert e < 5e-3
    lang = eng.stage(B, ncmp, T, "lang", torch.float32, (B, 1, 5, 256)).cpu()
    e = (lang - st["lang"]).abs().max().item()
    print("lang abs err", e)
    assert e < 5e-3
    ps = eng.stage(B, ncmp, T, "prop_seg", torch.float32, (B, 1, 200, 512)).cpu()
    e = (ps - st["prop_seg"]).abs().max().item()
    print("prop_seg abs err", e, "max", st["prop_seg"].abs().max().item())
    assert e < 1e-2
    oo = eng.stage(B, ncmp, T, "obj_outA", torch.float32, (B, 1, 200, 512)).cpu()
    e = (oo - st["obj_out"]).abs().max().item()
    print("obj_out abs err", e)
    assert e < 3e-2
    if fused:
        return        # the fused mul_tx tail runs lin2 + the score head itself: its output never leaves the chip
    # the last mul_tx layer writes only the 16-bit copy its consumer (the f16 score head) reads
    mo = eng.stage(B, ncmp, T, "mul_outA16", torch.float16, (40, 100, 768)).cpu().float()
    e = (mo - st["mul_out"]).abs().max().item()
    print("mul_out abs err", e)
    assert e < 3e-2


# ---- language encoder shared by a group of in-flight batches ---------------------------------
def _group_members(name, n):
    """n batches of the case's shape with different data seeds (member 0 = the golden case)."""
    out = []
    for k in range(n):
        key = name if k == 0 else f"{name}#m{k}"
        if k:
            c = dict(cases.CASES[name])
            c["dseed"] = c["dseed"] + 101 * k
            cases.CASES[key] = c
        try:
            out.append(cases.build(key))
        finally:
            if k:
                del cases.CASES[key]
    return out


@pytest.mark.parametrize("name,n", [("full/cfg2_vog_spat_gt5_bs4", 4), ("full/cfg2_ragged", 3), ("full/cfg5_vog_svsq_gt5_bs16", 2),
                                    ("small/vog_temp", 4)])
def test_batched_requests_match_standalone_forwards(name, n):
    """`make_batched`: n requests served as ONE forward (their rows back to back in one slot). Rows never interact: every
    member gets the outputs of its own stand-alone forward (same bound as the shared-language group: the fp32 summation order
    of a GEMM row does not depend on M, but sentences of another request change T = the longest sentence, hence the BiLSTM
    schedule of the padded steps), over graph replays."""
    from tests.gpu_util import engine_mod, comm_for
    members = _group_members(name, n)
    cfg, sd, _, c = members[0]
    eng = engine_mod.VogEngine(cfg, comm_for(c))
    eng.load_state_dict(sd)
    devs = [{k: torch.from_numpy(v).cuda() for k, v in m[2].items()} for m in members]
    refs = []
    for dv in devs:
        o = eng.forward(dv)
        refs.append({k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
    torch.cuda.synchronize()
    bt = eng.make_batched(devs, graph=True)
    ncmp = members[0][2]["new_srl_idxs"].shape[1]
    exact = True
    for rep in range(3):
        bt.big.out["mdl_outs"].fill_(float("nan"))
        outs = bt.launch()
        torch.cuda.synchronize()
        for m, (ref, out) in enumerate(zip(refs, outs)):
            e = (ref["mdl_outs"] - out["mdl_outs"]).abs().max().item()
            assert e <= 1.5e-3, (name, m, e)
            exact &= e == 0.0
            pa, pb = eng.unpack_pred(ref["pred_rec"], ncmp), eng.unpack_pred(out["pred_rec"].contiguous(), ncmp)
            assert (pa["scores"] - pb["scores"]).abs().max().item() <= 4e-4, (name, m)
    print(name, "batched == stand-alone bit for bit:", exact)
    # replay with the requests rotated by one position (update_member): outputs rotate with them - to the bound above, not bit
    # for bit: the tail kernels rotate their k order by the row block's position on its XCD (fp32 summation order is a
    # function of the position in the slot), 5e-4 on the logits at full size
    if all(int(d_["srl_arg_word_mask_len"].max()) <= bt.big.T for d_ in devs):
        first = [{k: v.clone() for k, v in o.items()} for o in outs]
        for m in range(n):
            bt.update_member(m, devs[(m + 1) % n])
        outs = bt.launch()
        torch.cuda.synchronize()
        worst = 0.0
        for m in range(n):
            worst = max(worst, (outs[m]["mdl_outs"] - first[(m + 1) % n]["mdl_outs"]).abs().max().item())
        print(name, "rotated requests: max logit difference", worst)
        assert worst <= 1.5e-3, (name, worst)


@pytest.mark.parametrize("name,n", [("full/cfg2_vog_spat_gt5_bs4", 4), ("full/cfg2_ragged", 4),
                                    ("full/vog_sep_gt5_bs4_ragged", 2), ("full/cfg3_vog_temp_gt5_bs8", 2),
                                    ("full/cfg1_igrnd_spat_gt5_bs2", 3), ("small/vog_spat", 4),
                                    ("small/vgrnd_sep", 2)])
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_group_language_encoder_matches_standalone_forwards(name, n, mode):
    """Every member of a group (one shared BiLSTM pass for all members' sentences) must get the
    outputs of its own stand-alone forward: rows never interact, only the fp32 summation order of
    the input-projection GEMMs differs (M = sum of rows picks another tile shape)."""
    from tests.gpu_util import engine_mod, comm_for
    members = _group_members(name, n)
    cfg, sd, _, c = members[0]
    eng = engine_mod.VogEngine(cfg, comm_for(c))
    eng.load_state_dict(sd)
    devs = [{k: torch.from_numpy(v).cuda() for k, v in m[2].items()} for m in members]
    refs = []
    for dv in devs:
        o = eng.forward(dv)
        refs.append({k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
    torch.cuda.synchronize()
    grp = eng.make_group(devs, graph=(mode == "graph"))
    for rep in range(2):                      # replay: state is re-zeroed inside the program
        for s in grp.slots:
            s.out["mdl_outs"].fill_(float("nan"))
        torch.cuda.synchronize()
        outs = grp.launch()
        torch.cuda.synchronize()
        ncmp = members[0][2]["new_srl_idxs"].shape[1]
        for m, (ref, out) in enumerate(zip(refs, outs)):
            # a 1-ulp fp32 difference in a gate can flip the 16-bit rounding of h; downstream that is
            # the same size of perturbation as the 16-bit plan itself (budget: tests/test_quant_budget.py),
            # so the bound is a fraction of the golden tolerances, not bit equality
            e = (ref["mdl_outs"] - out["mdl_outs"]).abs().max().item()
            assert e <= 1.5e-3, (name, mode, m, e)
            pa, pb = eng.unpack_pred(ref["pred_rec"], ncmp), eng.unpack_pred(out["pred_rec"], ncmp)
            assert (pa["scores"] - pb["scores"]).abs().max().item() <= 4e-4, (name, mode, m)
    # member 0 is the golden case: the reference-parity bound holds for the grouped path too
    g = np.load(cases.golden_path(name))
    pred = eng.unpack_pred(outs[0]["pred_rec"], ncmp)
    tol = (1e-3, 6e-3)
    _check_against(name, outs[0], pred, g, None, tol_rel=tol[0], tol_logit=tol[1])


class _OracleGolden(dict):
    """An oracle result in the shape `_check_against` expects from a golden .npz."""
    @property
    def files(self):
        return list(self.keys())


def _oracle_golden(cfg, sd, batch, c):
    o = oracle_run(cfg, sd, batch, c)
    keys = ("mdl_outs", "mdl_outs_eval", "scores", "boxes", "indexs", "vidf_outs", "fin_scores", "fin_scores_loss")
    return _OracleGolden({k: o[k].numpy() for k in keys if k in o})


def _variant(name, dseed_shift):
    """The inputs of case `name` re-drawn with another data seed (same weights, same shapes)."""
    key = f"{name}#v{dseed_shift}"
    cc = dict(cases.CASES[name])
    cc["dseed"] = cc["dseed"] + dseed_shift
    cases.CASES[key] = cc
    try:
        _, _, b, _ = cases.build(key)
    finally:
        del cases.CASES[key]
    return b


@pytest.mark.parametrize("name", ["full/cfg2_ragged", "full/cfg3_vog_temp_gt5_bs8"])
def test_slot_replay_with_new_inputs(name):
    """A captured slot replayed on CHANGED inputs: `update_inputs(B)` + launch, then a batch assembled on the
    device straight into the slot's buffers + launch, then A again. Every replay must equal the eager forward
    on the same data bit for bit (a stale hand-off slot, workspace stage or graph-baked pointer surviving from
    the previous replay would show here, not in replays of identical inputs) and the CPU oracle within the
    golden tolerances."""
    import importlib
    dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    from oracle import vog_oracle as vo
    eng, cfg, sd, batch_a, c, dev_a = build_engine(name)
    batch_b = _variant(name, 41)
    T = int(max(batch_a["srl_arg_word_mask_len"].max(), batch_b["srl_arg_word_mask_len"].max()))
    ncmp = batch_a["new_srl_idxs"].shape[1]
    slot = eng.make_slot(dev_a, T=T, graph=True)
    keys = ("mdl_outs", "mdl_outs_eval", "pred_rec")

    def check(batch, tag):
        out = slot.launch()
        torch.cuda.synchronize()
        got = {k: out[k].clone() for k in keys}
        ref = eng.forward({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in batch.items()}, T=T)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[k], ref[k]), (tag, k)
        g = _oracle_golden(cfg, sd, batch, c)
        _check_against(f"{name}/{tag}", got, eng.unpack_pred(got["pred_rec"], ncmp), g, None, tol_rel=1e-3, tol_logit=6e-3)
        return got

    first = check(batch_a, "A")
    slot.update_inputs({k: torch.from_numpy(v) for k, v in batch_b.items()})
    second = check(batch_b, "B")
    assert not torch.equal(first["mdl_outs"], second["mdl_outs"])
    # device-side batch assembly (csrc/assemble.hip) into the slot's input buffers
    conc = cfg.ds.conc_type
    B = batch_a["num_cmp_msk"].shape[0]
    it = synth.make_items(B, 4, c["nppf0"], seed=23)
    asm = dls.DeviceBatchAssembler(cfg, {"num_prop_per_frm": c["nppf0"]})
    asm({k: torch.from_numpy(v).cuda() for k, v in it.items()}, out={k: slot.inp[k] for k in dls.FWD_KEYS},
        with_loss_keys=False)
    torch.cuda.synchronize()
    batch_c = dict(batch_b)
    batch_c.update({k: slot.inp[k].cpu().numpy() for k in dls.FWD_KEYS})
    ref_asm = vo.assemble_batch(it, conc, 10, c["nppf0"])
    for k in dls.FWD_KEYS:
        assert np.array_equal(batch_c[k], ref_asm[k]), k
    third = check(batch_c, "assembled")
    assert not torch.equal(third["mdl_outs"], second["mdl_outs"])
    slot.update_inputs({k: torch.from_numpy(v) for k, v in batch_a.items()})
    again = check(batch_a, "A again")
    for k in keys:
        assert torch.equal(again[k], first[k]), k


@pytest.mark.parametrize("via", ["zero_copy", "dma_node", "device"])
@pytest.mark.parametrize("name", ["full/cfg2_ragged", "full/cfg3_vog_temp_gt5_bs8"])
def test_fed_slot_reads_the_batch_from_pinned_host_memory(name, via):
    """`Slot.feed_from` (vog_graph_capture_fed): the graph's first kernels read the per-video items and the word-level arrays
    of the batch from a pinned staging buffer (zero copy), assemble / copy them into the slot's inputs, then run the forward -
    one launch per step (`dma_node`: the graph starts with one transfer of the packed buffer and the kernels read the device
    copy; `device`: the caller uploads the packed buffer on a copy stream, the graph reads the device copy). Two different batches written into the SAME host buffer one after the other: every replay equals the
    eager forward on the device-assembled batch bit for bit; the assembled inputs equal the oracle's assembly."""
    import importlib
    dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    from oracle import vog_oracle as vo
    eng, cfg, sd, batch_a, c, dev_a = build_engine(name)
    batch_b = _variant(name, 41)
    T = int(max(batch_a["srl_arg_word_mask_len"].max(), batch_b["srl_arg_word_mask_len"].max()))
    slot = eng.make_slot(dev_a, T=T, graph=True)
    conc = cfg.ds.conc_type
    B = batch_a["num_cmp_msk"].shape[0]
    asm = dls.DeviceBatchAssembler(cfg, {"num_prop_per_frm": c["nppf0"]})
    lang_keys = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture",
                 "srl_arg_inds_msk", "num_cmp_msk")
    items = [synth.make_items(B, 4, c["nppf0"], seed=s) for s in (23, 29)]
    stg = dls.PackedStaging({**{k: np.zeros_like(items[0][k]) for k in dls.FWD_KEYS},
                             **{k: np.zeros_like(batch_a[k]) for k in lang_keys}})
    slot.feed_from(stg, asm, via=via)
    cs = torch.cuda.Stream()
    assert set(slot.fed_keys) == set(dls.FWD_KEYS) | set(lang_keys)
    keys = ("mdl_outs", "mdl_outs_eval", "pred_rec")
    outs = []
    for it, lang in zip(items, (batch_b, batch_a)):
        stg.fill({k: it[k] for k in dls.FWD_KEYS})
        stg.fill({k: lang[k] for k in lang_keys})
        if via == "device":
            stg.upload_on(cs)
        out = slot.launch()
        if via == "device":
            stg.release()
        slot.consumed().synchronize()
        got = {k: out[k].clone() for k in keys}
        ref_asm = vo.assemble_batch(it, conc, 10, c["nppf0"])
        full = dict(batch_a)
        full.update({k: lang[k] for k in lang_keys})
        full.update({k: ref_asm[k] for k in dls.FWD_KEYS})
        for k in dls.FWD_KEYS + lang_keys:
            assert np.array_equal(slot.inp[k].cpu().numpy(), full[k]), k
        ref = eng.forward({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in full.items()}, T=T)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(got[k], ref[k]), k
        assert torch.isfinite(got["mdl_outs"]).all()
        outs.append(got)
    assert not torch.equal(outs[0]["mdl_outs"], outs[1]["mdl_outs"])


def test_fed_pipeline_serves_a_stream_of_batches():
    """`engine.FedPipeline` (2 streams x 2 fed slots): 10 different batches written into the staging buffers in turn, one transfer
    + one launch each; every result equals the eager forward on the oracle-assembled batch bit for bit, in submission order."""
    import importlib
    dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    engine_mod = importlib.import_module("vognet-pytorch_amd.engine")
    from oracle import vog_oracle as vo
    name = "full/cfg2_ragged"
    eng, cfg, sd, batch_a, c, dev_a = build_engine(name)
    variants = [batch_a, _variant(name, 41), _variant(name, 7)]
    T = int(max(b["srl_arg_word_mask_len"].max() for b in variants))
    B = batch_a["num_cmp_msk"].shape[0]
    asm = dls.DeviceBatchAssembler(cfg, {"num_prop_per_frm": c["nppf0"]})
    lang_keys = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture",
                 "srl_arg_inds_msk", "num_cmp_msk")
    it0 = synth.make_items(B, 4, c["nppf0"], seed=1)
    spec = {**{k: np.zeros_like(it0[k]) for k in dls.FWD_KEYS}, **{k: np.zeros_like(batch_a[k]) for k in lang_keys}}
    pipe = engine_mod.FedPipeline(eng, dev_a, spec, asm, streams=2, slots_per_stream=2, T=T)
    assert len(pipe.slots) == 4 and pipe.slots[2].ws.data_ptr() == pipe.slots[0].ws.data_ptr()
    keys = ("mdl_outs", "mdl_outs_eval", "pred_rec")
    got, fulls = [], []
    for i in range(10):
        it = synth.make_items(B, 4, c["nppf0"], seed=100 + i)
        lang = variants[i % 3]
        st = pipe.next_staging()
        st.fill({k: it[k] for k in dls.FWD_KEYS})
        st.fill({k: lang[k] for k in lang_keys})
        sl = pipe.submit()
        with torch.cuda.stream(pipe.stream_of(sl)):            # collect on the slot's stream: the slot is reused 4 batches later
            got.append({k: sl.out[k].clone() for k in keys})
        full = dict(batch_a)
        full.update({k: lang[k] for k in lang_keys})
        full.update({k: vo.assemble_batch(it, cfg.ds.conc_type, 10, c["nppf0"])[k] for k in dls.FWD_KEYS})
        fulls.append(full)
    pipe.synchronize()
    torch.cuda.synchronize()
    for i, (g, full) in enumerate(zip(got, fulls)):
        ref = eng.forward({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in full.items()}, T=T)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(g[k], ref[k]), (i, k)
    assert not torch.equal(got[0]["mdl_outs"], got[1]["mdl_outs"])


def test_copy_segments_odd_lengths_and_host_sources():
    """vog_copy_segments: several byte ranges in one launch, lengths that are no multiple of 16, sources in pinned host memory
    and in device memory."""
    lib = L.load()
    rng = np.random.default_rng(5)
    sizes = [16, 4096, 100003, 7, 33]
    srcs = [torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)) for n in sizes]
    held = [t.pin_memory() if i % 2 == 0 else t.cuda() for i, t in enumerate(srcs)]
    dsts = [torch.zeros(n + 16, dtype=torch.uint8, device="cuda") for n in sizes]
    arr = (L.CopySeg * len(sizes))()
    for i, (h, d, n) in enumerate(zip(held, dsts, sizes)):
        arr[i].src, arr[i].dst, arr[i].bytes = h.data_ptr(), d.data_ptr(), n
    L.check(lib.vog_copy_segments(arr, len(sizes), L.stream_ptr()), "vog_copy_segments")
    torch.cuda.synchronize()
    for s_, d, n in zip(srcs, dsts, sizes):
        assert torch.equal(d[:n].cpu(), s_) and int(d[n:].sum()) == 0, n


def test_persistent_lstm_handoff_is_deterministic_under_load():
    """Four graphs in flight for 3000 launches, every slot ALTERNATING between two input batches
    (A / B / A ..., copied into its buffers on its own stream before each launch): every output stays
    bit-identical to the first result for the same inputs. The persistent BiLSTM hands h between 64
    workgroups through self-validating values in per-step slots that the forward's prologue re-arms; a
    stale, torn or timed-out hand-off - or anything left over from the previous replay's DIFFERENT data -
    would show up here as a difference or a NaN (scratch/stress_lstm.py is the long version)."""
    name = "full/cfg2_ragged"
    eng, cfg, sd, batch, c, dev = build_engine(name)
    slots, streams, inputs = [], [], []
    for s in range(4):
        ab = [_variant(name, 17 * s + 1), _variant(name, 17 * s + 9)]
        T = int(max(b["srl_arg_word_mask_len"].max() for b in ab))
        inputs.append([{k: torch.from_numpy(v).cuda() for k, v in b.items()} for b in ab])
        slots.append(eng.make_slot(inputs[s][0], T=T, graph=True))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    refs = []
    for sl, st, ab in zip(slots, streams, inputs):
        r = []
        for which in (0, 1):
            with torch.cuda.stream(st):
                sl.update_inputs(ab[which])
            sl.launch(st)
            torch.cuda.synchronize()
            r.append({k: v.clone() for k, v in sl.out.items() if isinstance(v, torch.Tensor)})
            assert torch.isfinite(r[-1]["mdl_outs"]).all()
        assert not torch.equal(r[0]["mdl_outs"], r[1]["mdl_outs"])
        refs.append(r)
    last = [1] * 4
    for i in range(3000):
        s = i % 4
        which = (i // 4) % 2
        with torch.cuda.stream(streams[s]):
            slots[s].update_inputs(inputs[s][which], check_lengths=False)       # (no host sync in the loop)
        slots[s].launch(streams[s])
        last[s] = which
        if (i + 1) % 500 == 0:
            torch.cuda.synchronize()
            for sl, ref, w in zip(slots, refs, last):
                for k in ref[w]:
                    assert torch.equal(sl.out[k], ref[w][k]), (i, k)



def test_slots_sharing_a_workspace_on_one_stream():
    """`make_slot(share_ws_with=...)` (bench.py --rotate-inputs: N input sets cycling through a stream's workspace): slots that
    share a workspace and are launched one after the other on ONE stream give exactly what slots with their own workspaces
    give, in any interleaving (every forward re-initialises the state it needs in its prologue)."""
    name = "full/cfg2_vog_spat_gt5_bs4"
    eng, cfg, sd, batch, c, dev = build_engine(name)
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    bs = [synth.make_batch("spat", 4, c["nppf0"], vocab_size=c["vocab"], seed=900 + i) for i in range(3)]
    T = max(int(b["srl_arg_word_mask_len"].max()) for b in bs)
    own = [eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, T=T, graph=True) for b in bs]
    sh = [eng.make_slot({k: torch.from_numpy(v) for k, v in bs[0].items()}, T=T, graph=True)]
    sh += [eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, T=T, graph=True, share_ws_with=sh[0]) for b in bs[1:]]
    assert sh[1].ws.data_ptr() == sh[0].ws.data_ptr() == sh[2].ws.data_ptr()
    st = torch.cuda.Stream()
    for s in own:
        s.launch(st)
    for order in ([0, 1, 2], [2, 0, 1, 1, 0, 2]):
        for i in order:
            sh[i].launch(st)
        st.synchronize()
        for i in range(3):
            for k in ("mdl_outs", "mdl_outs_eval", "pred_rec"):
                assert torch.equal(sh[i].out[k], own[i].out[k]), (order, i, k)


# ---- round 5: the fp32 path (checkpoints outside the f16 envelope) behind EVERY engine surface --------------------------------
def test_fp32_path_behind_batched_group_and_fed_surfaces():
    """`auto` on a checkpoint whose attention sharpness is outside the f16 AND the hi + lo envelope (full/cfg2_sharp48: 455): batched
    requests, a shared-language group and host-fed slots all hand out the fp32 path's results (the reference golden / the eager
    precise forward to 1e-5), not the 16-bit kernels' (1.5e-3 there)."""
    import importlib
    dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    from tests.gpu_util import comm_for
    from oracle import vog_oracle as vo
    name = "full/cfg2_sharp48"           # (round 6: past the hi + lo plan's envelope too - x 16 runs that plan now)
    members = _group_members(name, 2)
    cfg, sd, batch0, c = members[0]
    eng = engine_mod.VogEngine(cfg, comm_for(c))
    eng.load_state_dict(sd)
    assert eng.precise is not None and eng.plan == "f32" and eng.sharpness > engine_mod.SPLIT_SHARPNESS_MAX
    devs = [{k: torch.from_numpy(v).cuda() for k, v in m[2].items()} for m in members]
    refs = [{k: v.clone() for k, v in eng.forward(dv).items() if isinstance(v, torch.Tensor)} for dv in devs]
    torch.cuda.synchronize()
    g = np.load(cases.golden_path(name))
    ncmp = batch0["new_srl_idxs"].shape[1]
    _check_against(name, refs[0], eng.unpack_pred(refs[0]["pred_rec"], ncmp), g, None, tol_rel=5e-5, tol_logit=1e-4)
    for kind, unit in (("batched", eng.make_batched(devs, graph=True)), ("group", eng.make_group(devs, graph=True))):
        for rep in range(2):
            outs = unit.launch()
            torch.cuda.synchronize()
            for m, (ref, out) in enumerate(zip(refs, outs)):
                # (x 48: logits of hundreds of nats - another GEMM tile shape for the 2 x larger batch is another fp32 summation
                # order, and the softmax passes that on at full gain: 4.6e-5 measured on logits of O(10); 1e-5 at x 16 in round 5)
                e = (ref["mdl_outs"] - out["mdl_outs"]).abs().max().item()
                assert e <= 2e-4, (kind, m, e)
                assert torch.equal(ref["pred_rec"], out["pred_rec"].contiguous()) or \
                    (ref["pred_rec"] - out["pred_rec"]).abs().max().item() <= 2e-4, (kind, m)
    # host-fed slots: the graph's copies / assembly land in the slot's input buffers, the fp32 forward reads them behind it
    B = batch0["num_cmp_msk"].shape[0]
    asm = dls.DeviceBatchAssembler(cfg, {"num_prop_per_frm": c["nppf0"]})
    lang_keys = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture",
                 "srl_arg_inds_msk", "num_cmp_msk")
    it0 = synth.make_items(B, 4, c["nppf0"], seed=1)
    spec = {**{k: np.zeros_like(it0[k]) for k in dls.FWD_KEYS}, **{k: np.zeros_like(batch0[k]) for k in lang_keys}}
    T = int(batch0["srl_arg_word_mask_len"].max())
    pipe = engine_mod.FedPipeline(eng, devs[0], spec, asm, streams=2, slots_per_stream=1, T=T)
    got, fulls = [], []
    for i in range(3):
        it = synth.make_items(B, 4, c["nppf0"], seed=300 + i)
        st = pipe.next_staging()
        st.fill({k: it[k] for k in dls.FWD_KEYS})
        st.fill({k: batch0[k] for k in lang_keys})
        sl = pipe.submit()
        with torch.cuda.stream(pipe.stream_of(sl)):
            got.append(sl.out["mdl_outs"].clone())
        full = dict(batch0)
        full.update({k: vo.assemble_batch(it, cfg.ds.conc_type, 10, c["nppf0"])[k] for k in dls.FWD_KEYS})
        fulls.append(full)
    pipe.synchronize()
    torch.cuda.synchronize()
    for i, (gm, full) in enumerate(zip(got, fulls)):
        ref = eng.forward({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in full.items()}, T=T)
        torch.cuda.synchronize()
        assert (gm - ref["mdl_outs"]).abs().max().item() <= 2e-4, i

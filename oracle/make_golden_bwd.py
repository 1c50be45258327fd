"""Backward goldens of the score head + the last mul_tx encoder layer's tail, from AUTOGRAD THROUGH THE
REFERENCE (TEST INFRASTRUCTURE; build container only).

For every listed case: build the reference model class (oracle/ref_import.py), run its forward with autograd
enabled (eval mode: no dropout), the reference loss class on seeded targets (as oracle/make_golden_loss.py)
and `loss.backward()`; collect the gradients of every parameter on the path lin2 <- LayerNorm <- FFN <-
LayerNorm <- Wo of the LAST mul_tx layer (code/transformer_code.py:21-31, 73-81, 189-203; code/mdl_vog.py:224-230)
and of the tail's two inputs (the concatenated heads = input of `wo`; the layer input through the residual =
gradient at the first LayerNorm's input). Stored per tensor: L2 norm, sum, and the values at seeded sample
positions (all of them when the tensor has <= 4096 elements) in tests/golden/bwd__<case>.npz.

    python -m oracle.make_golden_bwd
"""
from __future__ import annotations

import importlib
import os

import numpy as np
import torch

from oracle import cases, ref_import
from oracle.make_golden_loss import targets_for

BWD_CASES = ["small/vog_spat", "small/vog_temp", "full/cfg2_vog_spat_gt5_bs4", "small/vog_sep_r64", "full/cfg5_vog_svsq_gt5_bs16",
             "small/igrnd_spat", "small/vgrnd_temp", "small/vgrnd_sep", "full/cfg1_igrnd_spat_gt5_bs2",
             "full/cfg2_ragged", "small/vog_sep_cmpmsk",      # sentences of different lengths at full size; masked-out videos
             "small/vog_spat_3layers", "small/vog_temp_objonefrm", "small/vog_spat_noobj", "small/vog_spat_norel",   # the model's knobs
             "full/vgrnd_spat_gt5_bs4", "full/vog_spat_gt5_bs4_3layers"]   # round 4: the other reported model kind / the 3-layer ablation at full size
N_SAMPLE = 4096


def bwd_path(name: str) -> str:
    return os.path.join(os.path.dirname(cases.golden_path(name)), "bwd__" + name.replace("/", "__") + ".npz")


def param_names(layer: int):
    p = f"mult_txf.encoder.layers.{layer}"
    return {"wo": f"{p}.selfattn.layer.wo.weight", "ln1g": f"{p}.selfattn.layernorm.weight",
            "ln1b": f"{p}.selfattn.layernorm.bias", "w1": f"{p}.feedforward.layer.linear1.weight",
            "b1": f"{p}.feedforward.layer.linear1.bias", "w2": f"{p}.feedforward.layer.linear2.weight",
            "b2": f"{p}.feedforward.layer.linear2.bias", "ln2g": f"{p}.feedforward.layernorm.weight",
            "ln2b": f"{p}.feedforward.layernorm.bias", "wl": "lin2.0.weight", "bl": "lin2.0.bias",
            "wl2": "lin2.2.weight", "bl2": "lin2.2.bias"}


def sample_index(n: int, seed: int = 12345) -> np.ndarray:
    if n <= N_SAMPLE:
        return np.arange(n, dtype=np.int64)
    return np.sort(np.random.default_rng(seed + n).choice(n, N_SAMPLE, replace=False)).astype(np.int64)


def pack(rec: dict, key: str, g: np.ndarray):
    flat = np.asarray(g, np.float32).reshape(-1)
    idx = sample_index(flat.size)
    rec[key + "__shape"] = np.array(g.shape, np.int64)
    rec[key + "__norm"] = np.array(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    rec[key + "__sum"] = np.array(flat.astype(np.float64).sum())
    rec[key + "__val"] = flat[idx]


def reference_grads(name: str):
    cfg, batch, c, tg = targets_for(name)
    _, sd, _, _ = cases.build(name)
    torch.set_num_threads(8)
    mdl = ref_import.build_model(cfg, c["vocab"], c["nppf0"], sd)
    for p in mdl.parameters():
        p.requires_grad_(True)
    has_mul = hasattr(mdl, "mult_txf")
    layer = len(mdl.mult_txf.encoder.layers) - 1 if has_mul else 0
    last = mdl.mult_txf.encoder.layers[layer] if has_mul else None
    cap = {}

    def keep(nm):
        def fn(_m, inp):
            cap[nm] = inp[0]
            inp[0].retain_grad()
        return fn
    def keep_out(nm):
        def fn(_m, _inp, out):
            if nm not in cap:                                                   # first call only
                cap[nm] = out
                out.retain_grad()
        return fn
    hs = []
    if has_mul:
        hs = [last.selfattn.layer.wo.register_forward_pre_hook(keep("attn")),       # the concatenated heads
              last.selfattn.layernorm.register_forward_pre_hook(keep("t")),         # x + attn Wo^T
              last.register_forward_pre_hook(keep("mul_in"))]                       # the layer input (both paths)
    has_obj = hasattr(mdl, "obj_txf") and cfg.mdl.name in ("vog", "vgrnd") and len(mdl.obj_txf.encoder.layers) > 0
    if has_obj:
        ol = mdl.obj_txf.encoder.layers[len(mdl.obj_txf.encoder.layers) - 1]
        hs += [ol.selfattn.layer.wo.register_forward_pre_hook(keep("obj_attn")),
               ol.selfattn.layernorm.register_forward_pre_hook(keep("obj_t")),
               ol.register_forward_pre_hook(keep("obj_in")),
               ol.register_forward_hook(keep_out("obj_out"))]
    hs += [mdl.srl_arg_words_out_enc.register_forward_hook(keep_out("lang_enc")),   # relu(linear), before the mask
           mdl.lstm_out_feat_proj.register_forward_hook(keep_out("lstm_proj")),     # first call: every time step
           mdl.prop_encoder.register_forward_hook(keep_out("prop_enc")),
           mdl.seg_encoder.register_forward_hook(keep_out("seg_enc"))]
    import mdl_conc_single as mcs  # noqa: reference modules
    import mdl_conc_sep as mcp
    ct = cfg.ds.conc_type
    cls = {"temp": mcs.LossB_TEMP, "spat": mcs.LossB_SPAT, "sep": mcp.LossB_SEP, "svsq": mcp.LossB_SEP}[ct]
    lf = cls(ref_import.ref_cfg(cfg), ref_import.Munch(num_prop_per_frm=c["nppf0"]))
    inp = {k: torch.from_numpy(v).clone() for k, v in {**batch, **tg}.items()}
    for k in ("pad_frm_mask", "pad_pnt_mask"):
        inp[k] = inp[k].to(torch.uint8)
    orig = torch.masked_select
    torch.masked_select = lambda x, m, *a, **k: orig(x, m.bool() if m.dtype == torch.uint8 else m, *a, **k)
    try:
        out = mdl(inp)
        res = lf(out, inp)
        res["loss"].backward()
    finally:
        torch.masked_select = orig
        for h in hs:
            h.remove()
    params = dict(mdl.named_parameters())
    grads = {k: params[n].grad.detach().numpy() for k, n in param_names(layer).items() if n in params and params[n].grad is not None}
    if "attn" in cap and cap["attn"].grad is not None:       # (ImgGrnd / VidGrnd have no mul_tx: these hooks never fire)
        grads["d_attn"] = cap["attn"].grad.detach().reshape(-1, cap["attn"].shape[-1]).numpy()
        grads["d_x"] = cap["t"].grad.detach().reshape(-1, cap["t"].shape[-1]).numpy()
    # round 3, second slice onwards: every parameter the loss reaches ("p:<name>") and the gradients at the
    # seams between the pieces of the backward ("d_<seam>", rows x features)
    for n, p_ in params.items():
        if p_.grad is not None:
            grads["p:" + n] = p_.grad.detach().numpy()
    for nm, t in cap.items():
        if nm in ("attn", "t") or t.grad is None:
            continue
        grads["d_" + nm] = t.grad.detach().reshape(-1, t.shape[-1]).numpy()
    return grads, float(res["loss"]), layer


def make(name: str):
    grads, loss, layer = reference_grads(name)
    rec = {"loss": np.array(loss, np.float32), "layer": np.array(layer, np.int64)}
    for k, g in grads.items():
        pack(rec, k, g)
    np.savez_compressed(bwd_path(name), **rec)
    print(f"{name:36s} loss {loss:.6f}  " + " ".join(f"{k}:{float(rec[k + '__norm']):.3e}" for k in grads if not k.startswith("p:")),
          f" {os.path.getsize(bwd_path(name)) / 1024:.0f} KB")


if __name__ == "__main__":
    if not ref_import.available():
        raise SystemExit("reference tree not present; goldens are generated in the build container")
    import sys
    pref = sys.argv[1] if len(sys.argv) > 1 else ""
    for n in BWD_CASES:
        if n.startswith(pref):
            make(n)

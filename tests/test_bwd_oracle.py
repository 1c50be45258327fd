"""CPU: the backward fixtures (autograd through the REFERENCE modules, oracle/make_golden_bwd.py) against
autograd through the oracle restatement (oracle/vog_oracle.py forward + loss_forward): pins the fixtures'
meaning - which tensors, which order, which loss - where the GPU box has no reference."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import make_golden_bwd as mgb
from oracle import vog_oracle as vo
from oracle.make_golden_loss import targets_for

CASES = list(mgb.BWD_CASES)


def check_fixture(g, key, got, tol=1e-3):
    """got (numpy) against the stored samples / norm of tensor `key`: max abs error <= tol * max |reference|."""
    flat = np.asarray(got, np.float32).reshape(-1)
    assert tuple(g[key + "__shape"]) == tuple(np.asarray(got).shape), key
    idx = mgb.sample_index(flat.size)
    ref = g[key + "__val"]
    scale = max(float(np.abs(ref).max()), 1e-12)
    err = float(np.abs(flat[idx] - ref).max())
    assert err <= tol * scale, (key, err, scale)
    nrm = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    assert abs(nrm - float(g[key + "__norm"])) <= tol * float(g[key + "__norm"]), (key, nrm)
    return err / scale


@pytest.mark.parametrize("name", CASES)
def test_oracle_autograd_equals_reference_autograd(name):
    cfg, batch, c, tg = targets_for(name)
    _, sd, _, _ = cases.build(name)
    g = np.load(mgb.bwd_path(name))
    layer = int(g["layer"])
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    sdt = {k: v.clone().requires_grad_(True) for k, v in vo.to_torch(sd).items()}
    inp = vo.to_torch({**batch, **tg})
    torch.set_num_threads(8)
    out = vo.forward(oc, sdt, inp, keep_stages=True)
    st = out["stages"]
    seams = {"mul_in": "mul_tail_x", "obj_in": "obj_tail_x", "obj_attn": "obj_tail_attn", "obj_t": "obj_tail_t",
             "obj_out": "obj_out_seq", "lang_enc": "lang_enc", "lstm_proj": "lstm_full_output", "prop_enc": "prop_enc",
             "seg_enc": "seg_enc"}
    for k in ["mul_tail_attn", "mul_tail_t"] + list(seams.values()):
        if k in st:
            st[k].retain_grad()
    res = vo.loss_forward(oc, out, inp, loss_lambda=float(cfg.loss.loss_lambda))
    assert abs(float(res["loss"]) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    res["loss"].backward()
    # bound, relative to each tensor's largest entry: 2e-4 for the small cases (measured 2e-7); 2e-3 at full size, where two
    # fp32 computations with different summation orders (the reference's fused nn.LSTM / nn.Linear vs the oracle's loops)
    # can put a ReLU pre-activation of ~1e-7 on either side of zero - one row's contribution to a weight gradient then
    # differs (measured 6e-4 on linear1.weight at cfg 5, 2e-7 at cfg 2)
    TOL = 2e-3 if name.startswith("full/") else 2e-4
    worst = 0.0
    for k, n in mgb.param_names(layer).items():
        if (k + "__shape") in g.files:                       # (ImgGrnd / VidGrnd: no mul_tx layer, lin2 only)
            worst = max(worst, check_fixture(g, k, sdt[n].grad.numpy(), tol=TOL))
    if "mul_tail_attn" in st:
        d = st["mul_tail_attn"].shape[-1]
        # (activation gradients at a seam, like the ones below: behind three layers one flipped ReLU moved a row of d_x by
        # 2.1e-3 of the tensor's largest entry at full size - fp32 summation order, reference vs oracle)
        worst = max(worst, check_fixture(g, "d_attn", st["mul_tail_attn"].grad.reshape(-1, d).numpy(), tol=10 * TOL if name.startswith("full/") else TOL))
        worst = max(worst, check_fixture(g, "d_x", st["mul_tail_t"].grad.reshape(-1, d).numpy(), tol=10 * TOL if name.startswith("full/") else TOL))
    # every parameter the loss reaches, and the gradients at the seams between the pieces of the backward
    n_par = 0
    for key in g.files:
        if key.startswith("p:") and key.endswith("__shape"):
            n = key[2:-len("__shape")]
            assert sdt[n].grad is not None, n
            worst = max(worst, check_fixture(g, "p:" + n, sdt[n].grad.numpy(), tol=TOL))
            n_par += 1
    assert n_par >= 29, n_par
    for seam, stage in seams.items():
        if ("d_" + seam + "__shape") in g.files:
            t = st[stage]
            # (a flipped ReLU moves ONE ROW of an activation gradient by a visible amount: 10 x the parameter bound)
            worst = max(worst, check_fixture(g, "d_" + seam, t.grad.reshape(-1, t.shape[-1]).numpy(), tol=10 * TOL))
    print(name, "worst relative error", worst, "parameters", n_par)

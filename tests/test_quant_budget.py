"""Tolerance budget of the 16-bit MFMA path, simulated on the CPU oracle.

Rounds the operands of every matrix contraction the HIP path runs in 16 bit
(bf16 in obj_tx/mul_tx, f16 in LSTM / encoders / score head; fp32 accumulate)
and checks the induced error on pred_scores stays inside the 1e-3 relative
bound of north_star with margin. This sizes the tolerance the GPU tests use."""
import torch

from oracle import cases
from oracle import vog_oracle as vo


def _scheme(tx, rest):
    m = {"tx": tx, "enc": rest, "lstm": rest, "head": rest}

    def q(scope, x):
        return x.to(m[scope.split(".")[0]]).to(torch.float32)
    return q


def test_budget_cfg2():
    cfg, sd, batch, c = cases.build("full/cfg2_vog_spat_gt5_bs4")
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    with torch.no_grad():
        ref = vo.pred_head(oc, vo.forward(oc, sdt, inp), inp)["scores"]
        for tx, bound in ((torch.bfloat16, 8e-4), (torch.float16, 3e-4)):
            got = vo.pred_head(oc, vo.forward(oc, sdt, inp, quant=_scheme(tx, torch.float16)), inp)["scores"]
            nz = ref > 0
            rel = ((got - ref).abs() / ref.clamp(min=1e-6))[nz].max().item()
            assert rel < bound, (tx, rel)

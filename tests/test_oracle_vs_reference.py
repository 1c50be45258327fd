"""Live comparison oracle <-> imported reference (build container only).

Skipped wherever /root/reference is absent (e.g. the GPU box)."""
import numpy as np
import pytest
import torch

from oracle import cases, ref_import
from oracle import vog_oracle as vo

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree absent")


@pytest.mark.parametrize("name", ["small/vog_spat", "small/vog_sep", "small/vgrnd_temp",
                                  "small/igrnd_svsq", "full/cfg2_vog_spat_gt5_bs4"])
def test_live(name):
    cfg, sd, batch, c = cases.build(name)
    ref, _ = ref_import.run_reference(cfg, c["vocab"], c["nppf0"], sd, batch)
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    inp = vo.to_torch(batch)
    with torch.no_grad():
        out = vo.forward(oc, vo.to_torch(sd), inp)
        out.update(vo.pred_head(oc, out, inp))
    for k, v in ref.items():
        np.testing.assert_allclose(out[k].float().numpy(), v.float().numpy(), atol=2e-5, rtol=0,
                                   err_msg=k)


def test_reference_mutates_mask_oracle_does_not():
    """the reference overwrites -1 -> 0 in srl_arg_word_mask."""
    cfg, sd, batch, c = cases.build("small/vog_spat")
    mdl = ref_import.build_model(cfg, c["vocab"], c["nppf0"], sd)
    inp = {k: torch.from_numpy(v).clone() for k, v in batch.items()}
    with torch.no_grad():
        mdl(inp)
    assert (inp["srl_arg_word_mask"] >= 0).all()
    assert (batch["srl_arg_word_mask"] < 0).any()

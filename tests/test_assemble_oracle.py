"""CPU: the oracle's SPAT / TEMP batch assembly (oracle/vog_oracle.py assemble_batch) bit-exact against
fixtures produced by the REFERENCE loader methods (oracle/make_golden_assemble.py); live in the build container."""
import numpy as np
import pytest

from oracle import cases, make_golden_assemble as mga, ref_import
from oracle import vog_oracle as vo


@pytest.mark.parametrize("conc", ["spat", "temp"])
def test_oracle_assembly_vs_reference_fixture(conc):
    it = mga.items()
    g = np.load(mga.path(conc))
    assert str(g["sha_items"]) == cases.digest(it), "item generator drifted"
    res = vo.assemble_batch(it, conc, 10, mga.SHAPE["nppf0"])
    for k in mga.KEYS:
        assert res[k].shape == g[k].shape, k
        assert np.array_equal(res[k], g[k].astype(res[k].dtype)), k
    assert g["num_box"].tolist()[1] == 0 and (g["pad_frm_mask"][1] == 1).all()      # the no-gt-box query


@pytest.mark.skipif(not ref_import.available(), reason="reference tree absent (GPU box)")
@pytest.mark.parametrize("conc", ["spat", "temp"])
def test_oracle_assembly_vs_reference_live(conc):
    it = mga.items()
    ref = mga.reference_assemble(it, conc)
    res = vo.assemble_batch(it, conc, 10, mga.SHAPE["nppf0"])
    for k in mga.KEYS:
        assert np.array_equal(res[k], ref[k].astype(res[k].dtype)), k

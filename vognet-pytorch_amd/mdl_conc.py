"""Concat-strategy mixins (reference code/mdl_conc_single.py:23-177,
code/mdl_conc_sep.py:13-217 — forward halves only). How the (video, frame,
proposal) axes fold into (sequence, token) axes per strategy is implemented in
csrc/forward.hip `make_geo`; the mixins pin `cfg.ds.conc_type` to the class and
carry the loss classes' names for the selector.
"""
from __future__ import annotations

from torch import nn


class ConcBase:
    conc_types = ()

    def set_args_conc(self):
        ct = self.cfg.ds.conc_type
        assert ct in self.conc_types, f"cfg.ds.conc_type={ct!r} not served by {type(self).__name__}"


class ConcTEMP(ConcBase):
    """4 videos concatenated in time: 1 model video, nfrm = ncmp*10, nppf = nppf0."""
    conc_types = ("temp",)


class ConcSPAT(ConcBase):
    """4 videos tiled in space: 1 model video, nfrm = 10, nppf = ncmp*nppf0."""
    conc_types = ("spat",)


class ConcSEP(ConcBase):
    """videos kept separate (+ pred_cmp head); svsq = single video."""
    conc_types = ("sep", "svsq")

    def set_args_conc(self):
        ConcBase.set_args_conc(self)
        self.nfrms = self.num_sampled_frm
        self.nppf = self.num_prop_per_frm


class _LossOutOfScope(nn.Module):
    """LossB_* (mdl_conc_single.py:180-433, mdl_conc_sep.py:220-447) is training /
    val-loss only and not on the forward->prediction path: SURVEY.md 8(f) rank 1
    ("next"). The class exists so `get_mdl_loss_eval` keeps its 3-key contract."""
    loss_keys = ["loss", "mdl_out_loss"]

    def __init__(self, cfg, comm):
        super().__init__()
        self.cfg, self.comm = cfg, comm

    def forward(self, out, inp):
        raise NotImplementedError(f"{type(self).__name__}: loss is outside the forward hot path (SURVEY.md 8(f))")


class LossB_TEMP(_LossOutOfScope):
    pass


class LossB_SPAT(_LossOutOfScope):
    pass


class LossB_SEP(_LossOutOfScope):
    pass

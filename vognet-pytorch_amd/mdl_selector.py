"""Plugin selector: (cfg.ds.conc_type, cfg.mdl.name) -> {mdl, loss, eval} classes.
Same contract as reference code/mdl_selector.py:26-69 (unknown key ->
NotImplementedError)."""
from __future__ import annotations

from . import mdl_vog as M
from .eval_vsrl_corr import EvaluatorSEP, EvaluatorSPAT, EvaluatorTEMP
from .mdl_conc import LossB_SEP, LossB_SPAT, LossB_TEMP

_TABLE = {
    "sep": ({"igrnd": M.ImgGrnd_SEP, "vgrnd": M.VidGrnd_SEP, "vog": M.VOG_SEP}, LossB_SEP, EvaluatorSEP),
    "temp": ({"igrnd": M.ImgGrnd_TEMP, "vgrnd": M.VidGrnd_TEMP, "vog": M.VOG_TEMP}, LossB_TEMP, EvaluatorTEMP),
    "spat": ({"igrnd": M.ImgGrnd_SPAT, "vgrnd": M.VidGrnd_SPAT, "vog": M.VOG_SPAT}, LossB_SPAT, EvaluatorSPAT),
}


def get_mdl_loss_eval(cfg):
    conc_type = "sep" if cfg.ds.conc_type == "svsq" else cfg.ds.conc_type
    if conc_type not in _TABLE:
        raise NotImplementedError(f"conc_type {cfg.ds.conc_type!r}")
    mdls, loss, evl = _TABLE[conc_type]
    if cfg.mdl.name not in mdls:
        raise NotImplementedError(f"mdl.name {cfg.mdl.name!r}")
    return {"mdl": mdls[cfg.mdl.name], "loss": loss, "eval": evl}

"""Base model contract of the plugin surface.

Same constructor contract and hook names as the reference `AnetBaseMdl`
(code/mdl_base.py:11-107): `cls(cfg=cfg, comm=comm)`, `set_args` reading the
same cfg/comm fields, `build_lang_model / build_vis_model / build_conc_model`
hooks, and a `state_dict()` with the reference's key names — but the module
tree holds parameters only; `forward` hands raw device pointers to
libvog_hip.so (engine.py). There is no torch compute on the path.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import synth
from .engine import VogEngine


class _Node(nn.Module):
    """Pure parameter container (never called)."""


def _register(root: nn.Module, dotted: str, value: torch.Tensor):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    m.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))


def normalize_key(k: str) -> str:
    """Reference checkpoint key -> the plain key: DDP `module.` prefix (utils/trn_utils.py:536-592),
    the inner `module.` of transformers trained with mdl.{obj,mul}_tx.use_ddp=True
    (`mult_txf.module.encoder...`, code/mdl_vog.py:441-445,577-578) and the legacy LayerNorm
    gamma / beta names (trn_utils.py:560-565)."""
    if k.startswith("module."):
        k = k[7:]
    for pre in ("mult_txf.", "obj_txf."):
        if k.startswith(pre + "module."):
            k = pre + k[len(pre) + 7:]
    if "layernorm" in k and k.endswith(".gamma"):
        k = k[:-6] + ".weight"
    if "layernorm" in k and k.endswith(".beta"):
        k = k[:-5] + ".bias"
    return k


class AnetBaseMdl(nn.Module):
    def __init__(self, cfg, comm):
        super().__init__()
        self.cfg = cfg
        if comm is not None:
            assert isinstance(comm, dict)
            self.comm = dict(comm)
        else:
            self.comm = {}
        self._engine = None
        self._weights_dirty = True
        self.set_args()
        self.after_init()

    def after_init(self):
        self.build_model()

    def build_model(self):
        self._init_sd = synth.init_state_dict(
            self.cfg, self.vocab_size, seed=int(torch.initial_seed() % (2 ** 31)))
        self.build_lang_model()
        self.build_vis_model()
        self.build_conc_model()
        del self._init_sd

    def _take(self, prefixes):
        for k in list(self._init_sd.keys()):
            if k.startswith(tuple(prefixes)):
                _register(self, k, torch.from_numpy(np.ascontiguousarray(self._init_sd.pop(k))))

    def set_args(self):
        """Same fields as reference mdl_base.py:32-75."""
        c = self.comm
        self.vocab_size = c["vocab_size"]
        self.detect_size = c["detect_size"]
        self.input_encoding_size = self.cfg.mdl.input_encoding_size
        self.rnn_size = self.cfg.mdl.rnn.rnn_size
        self.num_layers = self.cfg.mdl.rnn.num_layers
        self.drop_prob_lm = self.cfg.mdl.rnn.drop_prob_lm
        self.itod = c["itod"]
        self.num_sampled_frm = self.cfg.ds.num_sampled_frm
        self.num_prop_per_frm = c["num_prop_per_frm"]
        self.unk_idx = int(c["wtoi"]["UNK"])
        self.t_attn_size = self.cfg.ds.t_attn_size
        self.srl_arg_len = self.cfg.misc.srl_arg_length
        self.set_args_mdl()
        self.set_args_conc()

    def set_args_mdl(self):
        return

    def set_args_conc(self):
        return

    def build_lang_model(self):
        raise NotImplementedError

    def build_vis_model(self):
        raise NotImplementedError

    def build_conc_model(self):
        raise NotImplementedError

    # ---- weights -> engine ------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts reference checkpoints: optional `module.` prefix and legacy
        LayerNorm gamma/beta names (utils/trn_utils.py:536-592)."""
        sd = {normalize_key(k): v for k, v in state_dict.items()}
        r = super().load_state_dict(sd, strict=strict)
        self._weights_dirty = True
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._weights_dirty = True
        return r

    def _param_version(self):
        # In-place updates of a parameter (optimizer.step, p.copy_ / p.add_ under no_grad, nn.init.*) bump
        # Tensor._version; re-assigning p.data moves data_ptr. NOT seen: writes through `p.data` (p.data.copy_,
        # p.data.fill_ - `.data` is a detached alias with its own version counter): call `mark_dirty()` (or
        # `refresh_weights()`) after weight surgery of that kind.
        # (the parameter LIST is cached: walking the module tree cost 0.86 ms per forward, 4 x the forward's own host time;
        # whatever replaces parameter objects - _apply, load_state_dict - goes through _weights_dirty and drops the cache)
        plist = getattr(self, "_plist", None)
        self._pv_calls = getattr(self, "_pv_calls", 0) + 1
        if plist is None or self._weights_dirty or (self._pv_calls & 63) == 0:     # (re-walked every 64th call: a Parameter
            plist = self._plist = list(self.parameters())                           # object swapped by hand is noticed late, not never)
        return tuple((id(p), p._version, p.data_ptr()) for p in plist)

    def mark_dirty(self):
        """The parameters were edited in a way `_param_version` cannot see: re-upload at the next forward."""
        self._weights_dirty = True

    def refresh_weights(self) -> VogEngine:
        """Re-register every parameter with the engine now (invalidates captured slots: new weights_epoch)."""
        self._weights_dirty = True
        return self.engine()

    def engine(self) -> VogEngine:
        if self._engine is None:
            self._engine = VogEngine(self.cfg, self.comm)
        ver = self._param_version()
        if self._weights_dirty or ver != getattr(self, "_uploaded_version", None):
            self._engine.load_state_dict(self.state_dict())
            self._weights_dirty = False
            self._uploaded_version = ver
        return self._engine

    supports_T_hint = True

    def check_faults(self) -> None:
        """Raise VogError if a forward issued so far lost its BiLSTM hand-off (its outputs are NaN). Judges what has COMPLETED:
        call it after synchronising (`torch.cuda.synchronize()`), before trusting results; `forward` itself checks the forwards
        before it."""
        if self._engine is not None:
            self._engine.check()
            self._engine.check_logit_scale()      # (round 6: the observed attention-logit scale against the precision plan)

    def forward(self, inp: Dict[str, torch.Tensor], T: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """`forward(inp) -> {'mdl_outs', 'mdl_outs_eval'[, 'vidf_outs',
        'fin_scores_loss', 'fin_scores']}` as Conc{TEMP,SPAT,SEP}.forward
        (mdl_conc_single.py:68-127; mdl_conc_sep.py:131-217). Inputs are
        borrowed and NOT modified (the reference overwrites srl_arg_word_mask).
        Adds '_pred_rec': packed prediction records of the evaluator head.
        `T`: the longest sentence of the batch if the caller already knows it (from the HOST copy of
        `srl_arg_word_mask_len`); without it the length is read back from the device as in the reference
        (mdl_vog.py:257 `.max().item()`), which drains the stream once per batch."""
        out = self.engine().forward(inp, T=T, with_pred=True)     # (raises VogError if an EARLIER forward's hand-off stalled)
        res = {k: v for k, v in out.items() if not k.startswith("_") and k != "pred_rec"}
        res["_pred_rec"] = out["pred_rec"]
        return res

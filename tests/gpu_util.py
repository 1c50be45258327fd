"""Shared helpers of the -m gpu parity tests (product path = libvog_hip via ctypes)."""
import importlib

import numpy as np
import torch

from oracle import cases
from oracle import vog_oracle as vo

pkg = importlib.import_module("vognet-pytorch_amd")
L = importlib.import_module("vognet-pytorch_amd.lib")
engine_mod = importlib.import_module("vognet-pytorch_amd.engine")


def rnd16(x: torch.Tensor, dtype: str) -> torch.Tensor:
    t = torch.bfloat16 if dtype == "bf16" else torch.float16
    return x.to(t)


def t16(dtype: str):
    return torch.bfloat16 if dtype == "bf16" else torch.float16


def comm_for(c):
    return {"vocab_size": c["vocab"], "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1},
            "num_prop_per_frm": c["nppf0"]}


_ENGINES = {}
_DEFAULT_OPTIONS = {"lstm_persistent": 1, "lstm_inject_stall": 0, "fused_tail": 1, "fused_enc": 1, "pair_launches": 1, "pair_mask": 15,
                    "fused_ih": 1, "enc_lean": -1}


_CASES = {}


def _case(name):
    """cases.build(name) once per session (0.3-1 s of seeded numpy per call, on ~600 uses): the arrays are shared and must
    not be written to (no test does); cfg is copied because callers set cfg.hip fields."""
    import copy
    hit = _CASES.pop(name, None)
    if hit is None:
        hit = cases.build(name)
        if len(_CASES) >= 10:                       # (a full-size state dict is 177 MB of host memory)
            _CASES.pop(next(iter(_CASES)))
    _CASES[name] = hit                              # (most recently used last)
    cfg, sd, batch, c = hit
    return copy.deepcopy(cfg), dict(sd), dict(batch), dict(c)


def build_engine(name, tx_dtype=None, cached=False):
    """tx_dtype None: the package default (`auto`: f16 kernels inside their envelope, hi + lo operands / the fp32 path beyond;
    engine.py). cached=True (round 6: the launch-structure variant tests, which only flip context options on the same
    checkpoints): the engine of (case, tx_dtype) is built once per session - registering 44 M weights costs ~1 s, most of such a
    test - and handed out with every option back at its default and fresh device inputs; an engine that has seen a stall or
    whose plan was raised at run time is rebuilt."""
    key = (name, tx_dtype)
    hit = _ENGINES.get(key) if cached else None
    if hit is not None and hit[0].stalls == 0 and hit[0].plan == hit[5]:
        eng, cfg, sd, batch, c, _ = hit
        for k, v in _DEFAULT_OPTIONS.items():
            eng.set_option(k, v)
        eng._fault_seen = int(eng._fault[0])
    else:
        cfg, sd, batch, c = _case(name)
        if tx_dtype is not None:
            cfg.hip.tx_dtype = tx_dtype
        eng = engine_mod.VogEngine(cfg, comm_for(c))
        eng.load_state_dict(sd)
        if cached:
            if len(_ENGINES) >= 48:
                _ENGINES.pop(next(iter(_ENGINES)))
            _ENGINES[key] = (eng, cfg, sd, batch, c, eng.plan)
    dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    return eng, cfg, sd, batch, c, dev


def oracle_run(cfg, sd, batch, c, keep_stages=False):
    oc = vo.OracleCfg.from_cfg(cfg, c["vocab"], c["nppf0"])
    inp = vo.to_torch(batch)
    with torch.no_grad():
        out = vo.forward(oc, vo.to_torch(sd), inp, keep_stages=keep_stages)
        out.update(vo.pred_head(oc, out, inp))
    return out


def rel_err(a: np.ndarray, ref: np.ndarray, floor=1e-6):
    return np.abs(a - ref) / np.maximum(np.abs(ref), floor)

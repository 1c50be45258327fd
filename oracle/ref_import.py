"""Import the reference implementation IN THE BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE. `/root/reference` does not exist on the GPU box; nothing
on the `-m gpu` path, `smoke()` or `bench.py` may import this module. It is
used by `oracle/make_golden.py` (to generate the committed fixtures under
`tests/golden/`) and by `tests/test_oracle_vs_reference.py` (skipped when the
reference tree is absent).

Recipe: the reference needs `munch`, `fairseq`, `yacs`,
`fire`, `fastprogress`, tensorboard — none installed, no network. Each is
stubbed in `sys.modules` with the few names the imported files touch; no
reference source is copied or modified.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("VOG_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "code", "mdl_vog.py"))


class Munch(dict):
    """Minimal attribute dict standing in for munch.Munch."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _to_munch(d):
    if isinstance(d, dict):
        return Munch({k: _to_munch(v) for k, v in d.items()})
    return d


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("munch", Munch=Munch)
    fs = mod("fairseq")
    fs.utils = mod("fairseq.utils")
    yc = mod("yacs")
    yc.config = mod("yacs.config", CfgNode=Munch)
    mod("fire", Fire=lambda *a, **k: None)
    fp = mod("fastprogress", progress_bar=lambda it, **k: it,
             master_bar=lambda it, **k: it)
    fp.fastprogress = mod("fastprogress.fastprogress",
                          progress_bar=lambda it, **k: it,
                          master_bar=lambda it, **k: it)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        import torch.utils
        tb = mod("torch.utils.tensorboard", SummaryWriter=object)
        torch.utils.tensorboard = tb
    for p in ("code", "utils"):
        path = os.path.join(REF_ROOT, p)
        if path not in sys.path:
            sys.path.insert(0, path)
    _installed = True


def ref_cfg(cfg):
    """Our CfgNode -> attribute dict the reference modules read."""
    return _to_munch(cfg.to_dict())


def build_model(cfg, vocab_size: int, nppf0: int, state_dict=None):
    """Instantiate the reference model class selected exactly as
    code/mdl_selector.py:26-69 does, eval mode, optional weights."""
    install_stubs()
    import torch
    import mdl_vog  # noqa: reference module
    table = {
        ("sep", "igrnd"): "ImgGrnd_SEP", ("sep", "vgrnd"): "VidGrnd_SEP", ("sep", "vog"): "VOG_SEP",
        ("temp", "igrnd"): "ImgGrnd_TEMP", ("temp", "vgrnd"): "VidGrnd_TEMP", ("temp", "vog"): "VOG_TEMP",
        ("spat", "igrnd"): "ImgGrnd_SPAT", ("spat", "vgrnd"): "VidGrnd_SPAT", ("spat", "vog"): "VOG_SPAT",
    }
    ct = "sep" if cfg.ds.conc_type in ("sep", "svsq") else cfg.ds.conc_type
    cls = getattr(mdl_vog, table[(ct, cfg.mdl.name)])
    comm = {"vocab_size": vocab_size, "detect_size": 431, "itod": {},
            "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
    mdl = cls(cfg=ref_cfg(cfg), comm=comm)
    if state_dict is not None:
        sd = {k: (torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v)
              for k, v in state_dict.items()}
        missing, unexpected = mdl.load_state_dict(sd, strict=True)
    mdl.eval()
    return mdl


def build_evaluator(cfg, nppf0: int):
    """Reference Evaluator* with the dataset-reading `after_init` replaced
    (the stock one opens annotation csv/json: code/eval_fn_corr.py:54-67)."""
    install_stubs()
    import torch
    import eval_vsrl_corr as ev
    ct = "sep" if cfg.ds.conc_type in ("sep", "svsq") else cfg.ds.conc_type
    base = {"sep": ev.EvaluatorSEP, "temp": ev.EvaluatorTEMP, "spat": ev.EvaluatorSPAT}[ct]

    class _Ev(base):
        def after_init(self):
            self.num_sampled_frm = self.num_frms

    comm = Munch(num_prop_per_frm=nppf0)
    return _Ev(ref_cfg(cfg), comm, torch.device("cpu"))


def run_reference(cfg, vocab_size, nppf0, state_dict, batch, with_pred=True):
    """Reference forward (+ prediction head) on CPU, inputs cloned per call
    (the reference mutates `srl_arg_word_mask` in place, mdl_vog.py:80-82)."""
    import torch
    mdl = build_model(cfg, vocab_size, nppf0, state_dict)
    inp = {k: torch.from_numpy(v).clone() if not isinstance(v, torch.Tensor) else v.clone()
           for k, v in batch.items()}
    with torch.no_grad():
        out = mdl(inp)
        res = {k: v.clone() for k, v in out.items()}
        if with_pred:
            evl = build_evaluator(cfg, nppf0)
            inp2 = {k: torch.from_numpy(v).clone() if not isinstance(v, torch.Tensor) else v.clone()
                    for k, v in batch.items()}
            pr = evl.get_out_results_boxes(out, inp2)
            res.update({"boxes": pr["boxes"].contiguous(), "scores": pr["scores"].contiguous(),
                        "indexs": pr["indexs"].contiguous()})
    return res, mdl

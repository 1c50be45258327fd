"""CLI with the reference's shape (code/main_dist.py:90-163):

    python -m vognet_amd.main_dist <uid> [--dotted.cfg.key=value ...] [--local_rank=N]

`fire` is not installed, so the `uid --a.b=c` syntax is parsed here; overrides go
through `update_from_dict` (unknown key / type mismatch -> AssertionError, as
code/extended_config.py:69-78). The trainer (`Learner`, utils/trn_utils.py) and
the dataset loaders are out of scope (SURVEY.md 2.1 #11,#14): without the 530 GB
dataset the driver runs the selected model on synthetic batches
(`--only_val=True` semantics: forward + prediction head + cross-rank gather) and
prints one JSON line with throughput and output checksums.
"""
from __future__ import annotations

import json
import sys
import time
from typing import Any, Dict, List, Tuple

import torch

from . import dist as D
from . import synth
from .extended_config import (get_default_cfg, key_maps, num_prop_per_frm, post_proc_config,
                              update_from_dict)
from .mdl_selector import get_mdl_loss_eval


def parse_argv(argv: List[str]) -> Tuple[str, Dict[str, Any]]:
    if not argv or argv[0].startswith("--"):
        raise SystemExit("usage: main_dist.py <uid> [--a.b.c=value ...]")
    uid, kw = argv[0], {}
    i = 1
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected argument {a!r}")
        if "=" in a:
            k, v = a[2:].split("=", 1)
        else:
            k = a[2:]
            if i + 1 < len(argv) and not argv[i + 1].startswith("--"):
                v = argv[i + 1]
                i += 1
            else:
                v = "True"
        kw[k] = v
        i += 1
    return uid, kw


def learner_init(uid: str, cfg):
    """Builds comm -> model -> eval like reference main_dist.py:31-87 (data and
    Learner replaced by the synthetic driver)."""
    sel = get_mdl_loss_eval(cfg)
    comm = {"vocab_size": 5000, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1},
            "num_prop_per_frm": num_prop_per_frm(cfg)}
    mdl = sel["mdl"](cfg=cfg, comm=comm)
    device = torch.device("cuda", torch.cuda.current_device())
    evl = sel["eval"](cfg, comm, device)
    return mdl, evl, comm


def main_dist(uid: str, **kwargs):
    cfg = get_default_cfg()
    cfg.uid = uid
    cfg.cmd = list(sys.argv)
    n_batches = int(kwargs.pop("synthetic_batches", 20))
    if "local_rank" in kwargs:
        cfg.do_dist = True
        torch.cuda.set_device(int(kwargs["local_rank"]))
        D.init_from_env("nccl")
        D.synchronize()
    cfg.num_gpus = torch.cuda.device_count()
    cfg = update_from_dict(cfg, kwargs, key_maps)
    cfg = post_proc_config(cfg)
    cfg.freeze()
    mdl, evl, comm = learner_init(uid, cfg)
    if not (cfg.only_val or cfg.only_test):
        raise NotImplementedError(
            "training (Learner.fit) is outside the forward hot path; run with --only_val=True")
    rank, world = D.get_rank(), D.get_world_size()
    bs = cfg.train.bsv
    device = torch.device("cuda", torch.cuda.current_device())
    batches = []
    for i in range(n_batches):
        b = synth.make_batch(cfg.ds.conc_type, bs, comm["num_prop_per_frm"], vocab_size=comm["vocab_size"],
                             seed=1000 * i + rank)
        batches.append({k: torch.from_numpy(v) for k, v in b.items()})
    torch.cuda.synchronize()
    t0 = time.time()
    chk = 0.0
    nq = 0
    for b in batches:
        b = {k: v.to(device) for k, v in b.items()}
        with torch.no_grad():
            out = mdl(b)
        rec = D.all_gather_records(out["_pred_rec"])
        nq += rec.shape[0]
        chk += float(out["mdl_outs_eval"].sum())
    torch.cuda.synchronize()
    dt = time.time() - t0
    if D.is_main_process():
        print(json.dumps({"uid": uid, "world": world, "queries": nq, "seconds": dt,
                          "queries_per_s": nq / dt, "checksum": chk,
                          "mdl": cfg.mdl.name, "conc_type": cfg.ds.conc_type}))
    return


if __name__ == "__main__":
    _uid, _kw = parse_argv(sys.argv[1:])
    main_dist(_uid, **_kw)

"""CLI with the reference's shape (code/main_dist.py:90-163):

    python -m vognet_amd.main_dist <uid> [--dotted.cfg.key=value ...] [--local_rank=N]

`fire` is not installed, so the `uid --a.b=c` syntax is parsed here; overrides go
through `update_from_dict` (unknown key / type mismatch -> AssertionError, as
code/extended_config.py:69-78). `--only_val=True` / `--only_test=True` run the reference's validation flow
(code/main_dist.py:143-157 -> `Learner.validate`, utils/trn_utils.py:443-468): the selected evaluator is
called as `eval_fn(mdl, loss_fn, dl, dl_name, rank=..., pred_path=<tmp_path>/predictions/<uid>)`, i.e.
forward -> device loss -> prediction records -> ONE cross-rank exchange per 16 batches (dist.RecordRing) ->
rank 0 writes `<pred_path>/<dl_name>_0.pkl` and scores it when the annotation files of cfg.ds exist; the
loss and metric dicts are printed like the reference prints them, followed by one JSON line.
Without `only_val` / `only_test` it builds `trn_utils.Learner` and runs `learn.fit` (code/main_dist.py:31-87, 125 ->
utils/trn_utils.py:701-775): per epoch the device training step (`train.FP32Trainer`: fp32 forward -> loss -> backward -> gradient
all-reduce -> Adam) over the training batches, then the validation flow above on the inference model carrying the
new weights, and `<tmp_path>/models/<uid>.pth` in the reference's checkpoint layout; `--train.resume=True` loads
it back (model + optimizer), as `Learner.load_model_dict`.
The dataset readers are out of scope: without the 530 GB dataset the loaders are lists of
synthetic batches (`--synthetic_batches=N`, the last validation batch a query short like the tail batch of a
`drop_last=False` loader).
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path
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
    """Builds comm -> model / loss / eval like reference main_dist.py:31-87 (the data side and the trainer
    replaced by the synthetic loader below)."""
    sel = get_mdl_loss_eval(cfg)
    comm = {"vocab_size": 5000, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1},
            "num_prop_per_frm": num_prop_per_frm(cfg)}
    mdl = sel["mdl"](cfg=cfg, comm=comm)
    loss_fn = sel["loss"](cfg, comm)
    device = torch.device("cuda", torch.cuda.current_device())
    evl = sel["eval"](cfg, comm, device)
    return mdl, loss_fn, evl, comm


def synthetic_loader(cfg, comm, n_batches: int, rank: int, world: int, train: bool = False, pin: bool = True):
    """Validation batches with every key the forward, the loss and the evaluator read,
    this rank's contiguous share (dist.shard_indices = NewDistributedSampler, utils/trn_utils.py:127-156);
    the last batch is one query short (validation loaders keep the tail, trn_utils.py:200-203)."""
    import numpy as np
    bs = int(cfg.train.bs if train else cfg.train.bsv)
    ct = cfg.ds.conc_type
    out = []
    idx = list(D.shard_indices(n_batches, rank, world))
    if not train and (n_batches - 1) in idx:
        # shard_indices wraps around: the one short batch may not be the first a rank sees (a loader yields its tail last)
        idx = [i for i in idx if i != n_batches - 1] + [n_batches - 1] * idx.count(n_batches - 1)
    for i in idx:
        b = synth.make_batch(ct, bs, comm["num_prop_per_frm"], vocab_size=comm["vocab_size"], seed=1000 * i + (500000 if train else 0))
        b.update(synth.make_targets(b, ct, comm["num_prop_per_frm"], seed=i))
        ncmp = b["num_cmp_msk"].shape[1]
        rng = np.random.default_rng(77 + i)
        perm = np.stack([rng.permutation(ncmp) for _ in range(bs)]).astype(np.int64)
        b.update({"ann_idx": np.arange(i * bs, (i + 1) * bs, dtype=np.int64),
                  "sent_idx": np.arange(i * bs, (i + 1) * bs, dtype=np.int64),
                  "permute": perm, "permute_inv": np.argsort(perm, axis=1).astype(np.int64)})
        if i == n_batches - 1 and bs > 1 and not train:
            b = {k: v[: bs - 1] for k, v in b.items()}
        t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
        if pin and torch.cuda.is_available():
            # what DataLoader(pin_memory=True) hands over: the evaluator's `.to(device, non_blocking=True)` is then an
            # asynchronous copy at link speed (pageable tensors go through the runtime's bounce buffers: ~15 GB/s, 67 ms
            # of the 134 ms a 128-batch validation loop took)
            t = {k: v.pin_memory() for k, v in t.items()}
        out.append(t)
    return out


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
    mdl, loss_fn, evl, comm = learner_init(uid, cfg)
    rank, world = D.get_rank(), D.get_world_size()
    # Learner.init_log_dirs (utils/trn_utils.py:341-368): <data.path = cfg.misc.tmp_path>/predictions/<uid>
    pred_path = Path(cfg.misc.tmp_path) / "predictions" / uid
    if not (cfg.only_val or cfg.only_test):
        # learner_init + learn.fit (code/main_dist.py:31-87, 125)
        from .trn_utils import DataWrap, Learner
        data = DataWrap(path=cfg.misc.tmp_path, train_dl=synthetic_loader(cfg, comm, n_batches, rank, world, train=True),
                        valid_dl=synthetic_loader(cfg, comm, max(2, n_batches // 2), rank, world))
        learn = Learner(uid=uid, data=data, mdl=mdl, loss_fn=loss_fn, cfg=cfg, eval_fn=evl, comm=comm)
        t0 = time.time()
        hist = learn.fit(epochs=int(cfg.train.epochs), lr=float(cfg.train.lr))
        torch.cuda.synchronize()
        if D.is_main_process():
            print(json.dumps({"uid": uid, "world": world, "epochs": len(hist), "train_steps": learn.trainer.num_it,
                              "seconds": time.time() - t0, "model_file": str(learn.model_file), "history": hist}))
        return hist
    dl_name = "valid" if cfg.only_val else "test"
    dl = synthetic_loader(cfg, comm, n_batches, rank, world)
    nq_local = sum(int(b["num_cmp_msk"].shape[0]) for b in dl)
    if hasattr(mdl, "engine"):
        mdl.engine()                                  # register the weights (once per model, 0.6 s) outside the timed loop
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        val_loss, val_acc = evl(mdl, loss_fn, dl, dl_name, rank=rank, pred_path=pred_path)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if D.is_main_process():
        print(val_loss)                               # as the reference (main_dist.py:147-148)
        print(val_acc)
        fname = pred_path / f"{dl_name}_0.pkl"
        print(json.dumps({"uid": uid, "world": world, "queries": nq_local * world, "seconds": dt,
                          "queries_per_s": nq_local * world / dt, "mdl": cfg.mdl.name,
                          "conc_type": cfg.ds.conc_type, "dl_name": dl_name, "pred_file": str(fname),
                          "val_loss": {k: float(v) for k, v in val_loss.items()},
                          "val_acc": {k: float(v) for k, v in val_acc.items()}}))
    return val_loss, val_acc


if __name__ == "__main__":
    _uid, _kw = parse_argv(sys.argv[1:])
    main_dist(_uid, **_kw)

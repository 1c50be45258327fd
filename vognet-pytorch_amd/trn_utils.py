"""`Learner` with the reference's surface (utils/trn_utils.py:265-860) over the device training step:

    learn = Learner(uid, data, mdl, loss_fn, cfg, eval_fn, opt_fn=None, device=...)
    learn.fit(epochs, lr) | learn.validate(db) | learn.testing(db) | learn.save_model_dict() | learn.load_model_dict(...)

`data` carries `.path`, `.train_dl`, `.valid_dl`, `.test_dl` (the reference's DataWrap, utils/trn_utils.py:250-262). Files land where
the reference puts them (init_log_dirs :341-368): `<path>/txt_logs/<uid>.txt`, `<path>/models/<uid>.pth`,
`<path>/predictions/<uid>/<dl_name>_0.pkl`. The iteration itself is `train.FP32Trainer.step` (forward -> loss -> backward ->
gradient all-reduce -> Adam on the device); validation is the evaluator on the inference model (16-bit HIP forward) carrying the
trainer's current weights. `opt_fn` is accepted for signature compatibility: the optimizer is the reference's Adam(betas (0.9, 0.99))
(code/main_dist.py:55) as `vog_adam_f32`. Progress bars, tensorboard and the python logger are not reproduced.
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import dist as D
from .train import FP32Trainer, SmoothenDict


def DataWrap(path, train_dl=None, valid_dl=None, test_dl=None):
    """utils/trn_utils.py:250-262."""
    return SimpleNamespace(path=Path(path), train_dl=train_dl, valid_dl=valid_dl, test_dl=test_dl)


class Learner:
    def __init__(self, uid: str, data, mdl, loss_fn, cfg, eval_fn, opt_fn=None, device=None, comm=None, train_mode: bool = True):
        self.uid, self.data, self.mdl, self.loss_fn, self.cfg, self.eval_fn = uid, data, mdl, loss_fn, cfg, eval_fn
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.rank = D.get_rank()
        self.comm = comm if comm is not None else getattr(mdl, "comm", None)
        self.loss_keys, self.met_keys = list(loss_fn.loss_keys), list(eval_fn.met_keys)
        self.log_keys = (["epochs"] + [f"trn_{k}" for k in self.loss_keys] + [f"val_{k}" for k in self.loss_keys]
                         + [f"val_{k}" for k in self.met_keys])
        self.init_log_dirs()
        self.num_it, self.num_epoch, self.best_met = 0, 0, 0.0
        if bool(cfg.train.get("use_reduce_lr_plateau", False)):
            # the reference steps ReduceLROnPlateau on the validation metric (utils/trn_utils.py:470-483, 760-763); its
            # cfg.reduce_factor / cfg.patience are defined by no config - refuse instead of silently ignoring the flag
            raise NotImplementedError("cfg.train.use_reduce_lr_plateau: the plateau scheduler is not implemented (constant LR only)")
        # dropout masks: independent across ranks and runs (the reference draws from torch's per-process generator)
        base_seed = int(torch.initial_seed()) & 0x7FFFFFFF
        self.trainer = FP32Trainer(cfg, self.comm, mdl.state_dict(), loss_fn, lr=float(cfg.train.lr), dropout=train_mode,
                                   dropout_seed=(base_seed * D.get_world_size() + self.rank) & 0x7FFFFFFF)
        loaded_opt = False
        if cfg.train.resume:
            loaded_opt = bool(self.load_model_dict(resume_path=cfg.train.resume_path, load_opt=cfg.train.load_opt)) and bool(cfg.train.load_opt)
        # DistributedDataParallel broadcasts rank 0's parameters at construction: the replicas must not depend on seeding
        self.trainer.broadcast_from_rank0(with_optimizer=loaded_opt)
        if D.get_world_size() > 1:
            self._sync_model()

    # ---- files (init_log_dirs / create_log_dirs, utils/trn_utils.py:341-379)
    def init_log_dirs(self):
        p = Path(self.data.path)
        self.txt_log_file = p / "txt_logs" / f"{self.uid}.txt"
        self.model_file = p / "models" / f"{self.uid}.pth"
        self.predictions_dir = p / "predictions" / f"{self.uid}"
        if D.is_main_process():
            for d in (self.txt_log_file.parent, self.model_file.parent, self.predictions_dir):
                d.mkdir(parents=True, exist_ok=True)

    def update_log_file(self, towrite: str):
        if D.is_main_process():
            with self.txt_log_file.open("a") as f:
                f.write(towrite + "\n")

    @property
    def lr(self):
        return self.trainer.lr

    # ---- checkpoint (save_model_dict / load_model_dict, utils/trn_utils.py:533-630)
    def save_model_dict(self):
        if not D.is_main_process():
            return
        torch.save({"model_state_dict": {k: v.cpu() for k, v in self.trainer.state_dict().items()},
                    "optimizer_state_dict": self.trainer.optimizer_state_dict(), "num_it": self.trainer.num_it,
                    "num_epoch": self.num_epoch, "cfgtxt": json.dumps(self.cfg, default=str), "best_met": self.best_met},
                   self.model_file.open("wb"))

    def load_model_dict(self, resume_path: Optional[str] = None, load_opt: bool = False):
        mfile = self.model_file if not resume_path else Path(resume_path)
        if not mfile.exists():
            return False
        ck = torch.load(mfile.open("rb"), weights_only=False)
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ck["model_state_dict"].items()}
        self.trainer.params.update({k: v.to(self.trainer.dev, torch.float32).contiguous() for k, v in sd.items()
                                    if k in self.trainer.params})
        if load_opt and "optimizer_state_dict" in ck:
            self.trainer.load_optimizer_state_dict(ck["optimizer_state_dict"])
        # the Learner's iteration counter is restored on every load (utils/trn_utils.py:588-590), with or without the
        # optimizer state: it also numbers the dropout masks, which must not restart. Adam's own step (bias correction) is
        # NOT touched here: without `load_opt` the reference builds a fresh Adam at step 0 (m = v = 0), and a step count of N
        # against zero moments would switch the bias correction off (first update ~3.2 x lr); `load_optimizer_state_dict`
        # restores it together with m / v.
        if "num_it" in ck:
            self.trainer.num_it = self.num_it = int(ck["num_it"])
        self.num_epoch, self.best_met = int(ck.get("num_epoch", 0)), float(ck.get("best_met", 0.0))
        self._sync_model()
        return True

    def _sync_model(self):
        self.mdl.load_state_dict(self.trainer.state_dict(), strict=False)
        self.mdl.refresh_weights()

    # ---- loops
    def train_epoch(self, mb=None) -> Dict[str, float]:
        sm = SmoothenDict(self.loss_keys, 0.9)
        for batch in self.data.train_dl:
            batch = {k: v.to(self.trainer.dev) for k, v in batch.items()}
            sm.add_value(self.trainer.step(batch))
            self.num_it = self.trainer.num_it
        D.synchronize()
        out = dict(sm.smooth)
        if D.get_world_size() > 1:
            # reduce_dict(average=True) over the ranks (utils/trn_utils.py:61-90, 527-530)
            t = torch.tensor([float(out.get(k, 0.0)) for k in self.loss_keys], dtype=torch.float64, device=self.trainer.dev)
            torch.distributed.all_reduce(t)
            out = {k: float(v) / D.get_world_size() for k, v in zip(self.loss_keys, t.tolist())}
        return out

    def validate(self, db=None, mb=None, write_to_file: bool = False):
        if db is None:
            dl, dl_name = self.data.valid_dl, "valid"
        elif isinstance(db, dict):
            assert len(db) == 1
            dl_name, dl = next(iter(db.items()))
        else:
            dl, dl_name = db, "valid"
        self._sync_model()
        with torch.no_grad():
            out_loss, out_acc = self.eval_fn(self.mdl, self.loss_fn, dl, dl_name, rank=self.rank, pred_path=self.predictions_dir)
        D.synchronize()
        if write_to_file:
            self.update_log_file("  ".join([f"val_{k} {float(out_loss[k]):.4f}" for k in self.loss_keys]
                                           + [f"val_{k} {float(out_acc[k]):.4f}" for k in self.met_keys]))
        return out_loss, out_acc, {}

    def testing(self, db):
        db = db if isinstance(db, dict) else {"dl0": db}
        res = {}
        for dl_name, dl in db.items():
            out_loss, out_acc, _ = self.validate({dl_name: dl}, write_to_file=True)
            res[dl_name] = (out_loss, out_acc)
        return res

    def fit(self, epochs: int, lr: Optional[float] = None, params_opt_dict=None, log=print):
        if lr is not None:
            self.trainer.lr = float(lr)
        self.update_log_file("  ".join(self.log_keys) + "\n")
        hist, st, met = [], time.time(), None
        for _ in range(int(epochs)):
            self.num_epoch += 1
            trn = self.train_epoch()
            val_loss, val_acc, _ = self.validate(self.data.valid_dl)
            met = float(val_acc[self.met_keys[0]])
            rec = {"epochs": self.num_epoch, **{f"trn_{k}": trn[k] for k in self.loss_keys},
                   **{f"val_{k}": float(val_loss[k]) for k in self.loss_keys}, **{f"val_{k}": float(val_acc[k]) for k in self.met_keys}}
            hist.append(rec)
            if self.best_met < met or not self.model_file.exists():
                self.best_met = max(self.best_met, met)
                self.save_model_dict()
            line = "  ".join(f"{v:.4f}" if isinstance(v, float) else str(v) for v in rec.values())
            self.update_log_file(line)
            if D.is_main_process():
                log("  ".join(f"{k} {v:.4f}" if isinstance(v, float) else f"{k} {v}" for k, v in rec.items()))
            D.synchronize()
        self.update_log_file(f"epochs done {self.num_epoch}. Total time taken {time.time() - st: 0.4f}\n")
        return hist

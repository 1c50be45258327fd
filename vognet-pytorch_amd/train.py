"""the training step of the reference's `Learner.train_epoch` (utils/trn_utils.py:485-532) for the
VOGNet model on the device:

    out = mdl(batch); loss = loss_fn(out, batch); loss.backward(); optimizer.step()

as forward (fp32, activations kept at the seams) -> `LossB_*` (vog_loss_fwd) -> `LossB_*.backward` (vog_loss_bwd) ->
`visual_backward` + `language_backward` (csrc/backward.hip) -> gradient all-reduce over the ranks
(`dist.all_reduce_grads`, the DistributedDataParallel step of code/main_dist.py:72-85) -> Adam (vog_adam_f32; the
reference's `torch.optim.Adam(betas=(0.9, 0.99))`, code/main_dist.py:55). Everything is a C-ABI call into
libvog_hip.so; torch tensors are device containers.

This is the fp32 path that pins the MATH against autograd through the reference (every parameter gradient, three
Adam steps); it shares no kernel with the 16-bit inference forward and is not tuned (one GEMM per BiLSTM time step).
Covers ImgGrnd / VidGrnd / VOGNet with every conc_type. `dropout=True` is the reference's train mode (LSTMEncoder 0.1 / 0.1,
attn_drop on attention probabilities and sub-layer outputs) with masks from a counter-based generator on the device - the
reference's masks come from torch's generator, so a step is statistically, not bitwise, its step; with `dropout=False` a
step equals the reference's with the model in eval mode. For sep / svsq the verb head runs forward only:
the reference's `loss` excludes verb_loss (code/mdl_conc_sep.py:434-436).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import backward as BW
from . import lib as L
from .engine import model_desc_from_cfg


class FP32Trainer:
    def __init__(self, cfg, comm, state_dict: Dict[str, torch.Tensor], loss_fn, lr: Optional[float] = None,
                 betas=(0.9, 0.99), eps: float = 1e-8, device: str = "cuda", process_group=None, dropout: bool = False,
                 dropout_seed: int = 0, bf16_gemm: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("FP32Trainer needs a GPU (libvog_hip.so kernels; there is no CPU fallback)")
        self.lib = L.load()
        self.cfg, self.loss_fn = cfg, loss_fn
        d = model_desc_from_cfg(cfg, comm)
        if cfg.mdl.name not in ("vog", "vgrnd", "igrnd"):
            raise NotImplementedError(f"no device training step for mdl.name = {cfg.mdl.name}")
        self.desc = d
        self.dev = torch.device(device)
        self.params = {k: v.detach().to(self.dev, torch.float32).contiguous().clone() for k, v in state_dict.items()
                       if torch.is_tensor(v) and v.is_floating_point()}
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self.lr = float(cfg.train.lr if lr is None else lr)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        # two counters (round 5): `num_it` = the Learner's iteration count (numbers the dropout masks; restored from a
        # checkpoint on every resume, utils/trn_utils.py:588-590); `adam_step` = torch.optim.Adam's per-parameter `step`
        # (bias correction), which the reference restores ONLY with the optimizer state (`load_opt`, :594-605) - a resume
        # without it builds a fresh Adam at step 0 whose m / v are zero.
        self.num_it = 0
        self.adam_step = 0
        self.pg = process_group
        # train-mode dropout (`mdl.train()` in Learner.train_epoch): LSTMEncoder's 0.1 / 0.1 (utils/mdl_srl_utils.py:77), the
        # transformers' cfg.mdl.{obj,mul}_tx.attn_drop on attention probabilities and sub-layer outputs; masks come from a
        # counter-based generator seeded per step (csrc/backward.hip::drop_scale) - NOT torch's stream, so a step is
        # statistically, not bitwise, the reference's. Off: a step equals the reference's in eval mode.
        # mixed precision: the tile GEMMs round their operands to bf16 (fp32 accumulation, fp32 master weights, fp32 everything
        # else); off = the fp32 path that is pinned against autograd through the reference. A per-thread switch of the library
        # (vog_train_set_int), set at the top of every step.
        self.bf16_gemm = bool(bf16_gemm)
        self.dropout, self.dropout_seed = bool(dropout), int(dropout_seed)
        self.p_lstm = (0.1, 0.1)
        self.p_obj, self.p_mul = float(cfg.mdl.obj_tx.attn_drop), float(cfg.mdl.mul_tx.attn_drop)

    # ---- geometry of one batch (Conc{TEMP,SPAT}: code/mdl_conc_single.py:24-37, 131-143)
    def _geo(self, batch):
        d = self.desc
        B = batch["srl_arg_words_ind"].shape[0]
        ncmp = batch["num_cmp_msk"].shape[1]
        nc_v = 1
        if self.cfg.ds.conc_type == "temp":
            nfrm, nppf = ncmp * d.nfrm0, d.nppf0
        elif self.cfg.ds.conc_type == "spat":
            nfrm, nppf = d.nfrm0, ncmp * d.nppf0
        else:                                           # sep / svsq: every video is its own sequence set (mdl_conc_sep.py:14-26)
            nc_v, nfrm, nppf = ncmp, d.nfrm0, d.nppf0
        name = self.cfg.mdl.name                       # ImgGrnd: encoders + lin2; VidGrnd: + obj_tx; VOGNet: + mul_tx (mdl_vog.py:286-744)
        has_obj = name == "vgrnd" or (name == "vog" and d.obj_to_use)
        seed = self._step_seed()
        dr = dict(drop_obj=(self.p_obj, seed), drop_mul=(self.p_mul, seed), drop_lang=self.p_lstm + (seed,)) if self.dropout else {}
        return dict(**dr, B=B, nc_v=nc_v, nfrm=nfrm, nppf=nppf, nsrl=d.nsrl, nppf0=d.nppf0, mul_layers=d.mul_layers if name == "vog" else 0,
                    mul_heads=d.mul_heads, mul_use_rel=bool(d.mul_use_rel), obj_layers=d.obj_layers if has_obj else 0, obj_heads=d.obj_heads,
                    obj_use_rel=bool(d.obj_use_rel), obj_one_frm=bool(d.obj_one_frm), vid_w=d.vid_w, vid_h=d.vid_h)

    def _step_seed(self) -> int:
        """One seed per optimisation step (forward and backward of a step recompute the same masks from it)."""
        return (self.dropout_seed * 1000003 + self.num_it + 1) & 0x7FFFFFFFFFFFFFFF

    def _stack_forward(self, stack, n_layers, pe_name, x, S, N, n, heads, boxes, drop=None):
        """-> (output, kept = (layer inputs, concatenated heads) for the backward)."""
        y, xs, cats = BW.stack_forward(self.params, stack, n_layers, pe_name, x, S, N, n, heads, boxes, drop=drop)
        return y, (xs, cats)

    def forward(self, batch, T=None):
        """fp32 forward on the device -> ({'mdl_outs' [B, nc_v, nsrl, NP] (, 'vidf_outs' [B, ncmp])}, activations at the seams of
        the backward, geometry). `T`: the longest sentence if the caller knows it (a slot does): no host read of the lengths."""
        g = self._geo(batch)
        p, lib, st = self.params, self.lib, L.stream_ptr()
        B, nc_v, nfrm, nppf, nsrl = g["B"], g["nc_v"], g["nfrm"], g["nppf"], g["nsrl"]
        NP = nfrm * nppf
        BV = B * nc_v                                                   # model "videos" (sequence sets)
        T = int(batch["srl_arg_word_mask_len"].max()) if T is None else int(T)
        lf = BW.language_backward(p, batch, T, self.desc.rnn_layers, drop=g.get("drop_lang"))
        lang = lf["_lang_enc"]
        f32 = lambda k: batch[k].to(self.dev, torch.float32)
        prop_feat = f32("pad_region_feature").reshape(BV * NP, -1).contiguous()
        seg_feat = f32("seg_feature_for_frms").reshape(-1, batch["seg_feature_for_frms"].shape[-1]).contiguous()
        props = f32("pad_proposals").reshape(BV * NP, -1).contiguous()
        pe = BW.linear_f32(prop_feat, p["prop_encoder.0.weight"], p["prop_encoder.0.bias"], True)["y"]
        se = BW.linear_f32(seg_feat, p["seg_encoder.0.weight"], p["seg_encoder.0.bias"], True)["y"]
        assert seg_feat.shape[0] * g["nppf0"] == BV * NP
        obj_x = torch.empty(BV * NP, pe.shape[1] + se.shape[1], dtype=torch.float32, device=self.dev)
        L.check(lib.vog_concat_rows_f32(L.ptr(pe), pe.shape[1], 1, L.ptr(se), se.shape[1], g["nppf0"], L.ptr(obj_x), BV * NP, st),
                "vog_concat_rows_f32")
        obj_out, obj_kept, mul_kept = obj_x, None, None
        if g["obj_layers"] > 0:
            if g["obj_one_frm"]:
                S, N, fdiv = BV * nfrm, nppf, float(nfrm)
            else:
                S, N, fdiv = BV, NP, 1.0
            ob = BW._Boxes(props, g["vid_w"], g["vid_h"], fdiv) if g["obj_use_rel"] else None
            obj_out, obj_kept = self._stack_forward("obj_txf", g["obj_layers"], "pe_obj_sub_enc.0", obj_x, S, N, N, g["obj_heads"], ob,
                                                    drop=g.get("drop_obj"))
        msk = batch["srl_arg_inds_msk"].to(self.dev, torch.int64).contiguous()
        nv = msk.shape[1]
        assert nv in (1, nc_v), "language axis does not match conc_type"
        lang_per_vid = 1 if (nv == nc_v and nc_v > 1) else 0
        dobj, dlang = obj_out.shape[1], lang.shape[1]
        # the [vis | lang] tokens: regrouped per frame for mul_tx, in (video, arg, proposal) order when lin2 reads them directly
        hf, hp = (nfrm, nppf) if g["mul_layers"] > 0 else (1, NP)
        mul_x = torch.empty(BV * nfrm * nsrl * nppf, dobj + dlang, dtype=torch.float32, device=self.dev)
        L.check(lib.vog_conc_f32_fwd(L.ptr(obj_out), L.ptr(lang), L.ptr(msk), L.ptr(mul_x), B, nc_v, hf, hp, nsrl, dobj, dlang,
                                     lang_per_vid, st), "vog_conc_f32_fwd")
        y = mul_x
        if g["mul_layers"] > 0:
            mb = BW._Boxes(props, g["vid_w"], g["vid_h"], float(nfrm)) if g["mul_use_rel"] else None
            y, mul_kept = self._stack_forward("mult_txf", g["mul_layers"], "pe_mul_sub_enc.0", mul_x, BV * nfrm, nsrl * nppf, nppf,
                                              g["mul_heads"], mb, drop=g.get("drop_mul"))
        M, dm = y.shape
        dhead = p["lin2.0.weight"].shape[0]
        scratch = torch.empty(M * dhead, dtype=torch.float32, device=self.dev)
        outs = torch.empty(B, nc_v, nsrl, NP, dtype=torch.float32, device=self.dev)
        L.check(lib.vog_score_head_f32(L.ptr(y), L.ptr(p["lin2.0.weight"]), L.ptr(p["lin2.0.bias"]), L.ptr(p["lin2.2.weight"]),
                                       L.ptr(p["lin2.2.bias"]), L.ptr(outs), L.ptr(scratch), scratch.numel() * 4, M, dm, dhead, BV,
                                       hf, hp, nsrl, st), "vog_score_head_f32")
        out = {"mdl_outs": outs}
        if self.cfg.ds.conc_type in ("sep", "svsq"):
            # verb head (mdl_conc_sep.py:64-129): reported by LossB_SEP as verb_loss; the reference's `loss` does not include
            # it (`out_loss = mdl_out_loss`, mdl_conc_sep.py:434-436), so nothing flows back through it
            hid = lf["_hid"]                                            # [B*nv, D]
            Fv = seg_feat.shape[0] // BV
            seg_mean = torch.empty(BV, se.shape[1], dtype=torch.float32, device=self.dev)
            L.check(lib.vog_row_mean_f32(L.ptr(se), L.ptr(seg_mean), BV, Fv, se.shape[1], st), "vog_row_mean_f32")
            sv = torch.empty(BV, hid.shape[1] + se.shape[1], dtype=torch.float32, device=self.dev)
            L.check(lib.vog_concat_rows_f32(L.ptr(hid), hid.shape[1], 1 if nv == nc_v else nc_v, L.ptr(seg_mean), se.shape[1], 1,
                                            L.ptr(sv), BV, st), "vog_concat_rows_f32")
            h1 = BW.linear_f32(sv, p["seg_verb_classf.0.weight"], p["seg_verb_classf.0.bias"], True)["y"]
            out["vidf_outs"] = BW.linear_f32(h1, p["seg_verb_classf.2.weight"], p["seg_verb_classf.2.bias"], False)["y"].reshape(B, nc_v)
        # kept for the backward: the inputs and concatenated heads of every encoder layer, the language side's whole scratch
        acts = {"mul_x": mul_x, "obj_x": obj_x, "prop_feat": prop_feat, "seg_feat": seg_feat, "props": props, "inds_msk": msk, "T": T,
                "mul_kept": mul_kept, "obj_kept": obj_kept, "lang_scratch": lf["_scratch"], "hid": lf["_hid"]}
        return out, acts, g

    def gradients(self, batch, exchange: bool = False):
        """-> (loss dict, {parameter name: gradient}) of one batch (no update). exchange: average the gradients over the
        ranks - the visual side's buckets are in flight while the language side's backward runs."""
        L.check(self.lib.vog_train_set_int(b"bf16_gemm", 1 if self.bf16_gemm else 0), "vog_train_set_int")
        try:
            return self._gradients(batch, exchange)
        finally:                                            # (process-wide switch: never leave it on behind an exception)
            self.lib.vog_train_set_int(b"bf16_gemm", 0)

    def _gradients(self, batch, exchange):
        out, acts, g = self.forward(batch)
        ld = self.loss_fn(out, batch)
        d_outs = self.loss_fn.backward(ld)
        grads = BW.visual_backward(self.params, g, acts, d_outs)
        gv = {k: v for k, v in grads.items() if not k.startswith("_")}
        fin_v = None
        if exchange:
            from .dist import all_reduce_grads_begin
            fin_v = all_reduce_grads_begin(gv)
        lg = BW.language_backward(self.params, batch, acts["T"], self.desc.rnn_layers, d_lang_enc=grads["_d_lang"], drop=g.get("drop_lang"),
                                  forward_scratch=acts["lang_scratch"])
        gl = {k: v for k, v in lg.items() if not k.startswith("_")}
        if exchange:
            fin_l = all_reduce_grads_begin(gl)
            fin_v()
            fin_l()
        gv.update(gl)
        return ld, gv

    def step(self, batch):
        """One `train_epoch` iteration: forward, loss, backward, (all-reduce,) Adam. -> the loss dict."""
        multi = self.pg is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                        and torch.distributed.get_world_size() > 1)
        ld, grads = self.gradients(batch, exchange=multi)
        self.num_it += 1
        self.adam_step += 1
        st = L.stream_ptr()
        for k in sorted(grads):
            p = self.params[k]
            if k not in self.m:
                self.m[k], self.v[k] = torch.zeros_like(p), torch.zeros_like(p)
            gk = grads[k].contiguous()
            L.check(self.lib.vog_adam_f32(L.ptr(p), L.ptr(gk), L.ptr(self.m[k]), L.ptr(self.v[k]), p.numel(), self.lr, self.betas[0],
                                          self.betas[1], self.eps, self.adam_step, st), "vog_adam_f32")
        return ld

    def broadcast_from_rank0(self, with_optimizer: bool = False) -> None:
        """What DistributedDataParallel does at construction (code/main_dist.py:72-85): every rank starts from rank 0's
        parameters (and Adam state, after a resume with load_opt) - replicas seeded differently would otherwise apply the
        averaged gradient to different weights. Name order; a no-op without a multi-rank group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1):
            return
        for k in sorted(self.params):
            dist.broadcast(self.params[k], src=0, group=self.pg)
        if with_optimizer:
            for k in sorted(self.params):
                if k not in self.m:
                    self.m[k], self.v[k] = torch.zeros_like(self.params[k]), torch.zeros_like(self.params[k])
                dist.broadcast(self.m[k], src=0, group=self.pg)
                dist.broadcast(self.v[k], src=0, group=self.pg)
            t = torch.tensor([self.num_it, self.adam_step], dtype=torch.int64, device=self.dev)
            dist.broadcast(t, src=0, group=self.pg)
            self.num_it, self.adam_step = int(t[0].item()), int(t[1].item())

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The parameters under the reference's key names (load into the inference model with `load_state_dict`)."""
        return {k: v.clone() for k, v in self.params.items()}

    # ---- checkpoint in the reference's layout (Learner.save_model_dict / load_model_dict, utils/trn_utils.py:533-630)
    def optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() layout (parameters numbered in state-dict key order)."""
        keys = list(self.params)
        state = {i: {"step": torch.tensor(float(self.adam_step)), "exp_avg": self.m[k].clone(), "exp_avg_sq": self.v[k].clone()}
                 for i, k in enumerate(keys) if k in self.m}
        return {"state": state, "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0,
                                                  "amsgrad": False, "params": list(range(len(keys)))}]}

    def load_optimizer_state_dict(self, osd):
        """Adam state by POSITION in state-dict key order - torch.optim.Adam numbers `mdl.parameters()`, which for the reference's
        modules is the order of `state_dict()` restricted to parameters. A state whose shape is not its parameter's (a
        checkpoint of another mdl.name / conc_type, or of a model with other sizes) is refused: vog_adam_f32 walks m / v with
        the parameter's element count."""
        keys = list(self.params)
        new_m, new_v, adam_step = {}, {}, self.adam_step
        for i, stt in osd["state"].items():
            if not 0 <= int(i) < len(keys):
                raise ValueError(f"optimizer state for parameter #{i}, the model has {len(keys)}")
            k = keys[int(i)]
            for nm in ("exp_avg", "exp_avg_sq"):
                if tuple(stt[nm].shape) != tuple(self.params[k].shape):
                    raise ValueError(f"optimizer state #{i} ({nm}) has shape {tuple(stt[nm].shape)}, parameter {k} has "
                                     f"{tuple(self.params[k].shape)}: this checkpoint belongs to another model")
            new_m[k] = stt["exp_avg"].to(self.dev, torch.float32).contiguous().clone()
            new_v[k] = stt["exp_avg_sq"].to(self.dev, torch.float32).contiguous().clone()
            adam_step = int(stt["step"])
        self.m.update(new_m)
        self.v.update(new_v)
        self.adam_step = adam_step
        g = osd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])


class SmoothenDict:
    """Exponentially smoothed loss values with bias correction (utils/trn_utils.py:228-262: SmoothenValue / SmoothenDict)."""

    def __init__(self, keys, beta: float):
        self.keys, self.beta, self.n = list(keys), beta, 0
        self.mov = {k: 0.0 for k in self.keys}
        self.smooth = {k: 0.0 for k in self.keys}

    def add_value(self, d):
        self.n += 1
        for k in self.keys:
            self.mov[k] = self.beta * self.mov[k] + (1 - self.beta) * float(d[k])
            self.smooth[k] = self.mov[k] / (1 - self.beta ** self.n)

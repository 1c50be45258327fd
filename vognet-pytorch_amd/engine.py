"""Host-side driver of libvog_hip for one model instance.

`VogEngine` owns a `vog_ctx` (weights registered under the reference's
state-dict key names), per-shape workspaces and captured hipGraphs, and turns a
batch dict of device tensors into the output dict of the reference `forward`
(+ the packed prediction records of the evaluator head). torch tensors are
containers only; every FLOP of the path runs in libvog_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import os

import numpy as np
import torch

from . import lib as L

NSRL_KEYS_I64 = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len",
                 "srl_arg_words_capture", "srl_arg_inds_msk", "num_cmp_msk")
F32_KEYS = ("pad_region_feature", "seg_feature_for_frms", "pad_proposals")


def model_desc_from_cfg(cfg, comm) -> L.ModelDesc:
    m = cfg.mdl
    hip = cfg.get("hip", {}) if hasattr(cfg, "get") else {}
    tx = hip.get("tx_dtype", "auto") if hasattr(hip, "get") else "auto"
    if tx in ("auto", "f32", "split"):
        # f16 transformers by default (round 5): same MFMA rate as bf16 on gfx950, three more mantissa bits. bf16 holds the
        # 1e-3 bound only while attention is near-uniform (random-init weights; fails at wq / wk x 8, logit std ~ 1); f16 holds
        # to x 12 ("envelope"). Past that `auto` routes forwards through the fp32 kernels
        # (`precise.PreciseForward`, decided per checkpoint in `load_state_dict` from `attention_sharpness`); "f32" forces it.
        # bf16 stays selectable (`cfg.hip.tx_dtype = "bf16"`: what BASELINE.json's config 2 names, what bench.py times).
        tx = "f16"
    d = L.ModelDesc()
    d.mdl_kind = L.MDL_KIND[m.name]
    d.conc_type = L.CONC_TYPE[cfg.ds.conc_type]
    d.vocab_size = int(comm["vocab_size"])
    d.emb_dim = m.input_encoding_size
    d.rnn_size = m.rnn.rnn_size
    d.rnn_layers = m.rnn.num_layers
    d.prop_dim, d.seg_dim = m.prop_feat_dim, m.seg_feat_dim
    d.prop_enc, d.seg_enc, d.lang_enc = (m.vsrl.prop_encode_size, m.vsrl.seg_encode_size,
                                         m.vsrl.lang_encode_size)
    d.obj_layers, d.obj_heads = m.obj_tx.n_layers, m.obj_tx.n_heads
    d.obj_use_rel, d.obj_one_frm, d.obj_to_use = (int(m.obj_tx.use_rel), int(m.obj_tx.one_frm),
                                                  int(m.obj_tx.to_use))
    d.mul_layers, d.mul_heads, d.mul_use_rel = (m.mul_tx.n_layers, m.mul_tx.n_heads,
                                                int(m.mul_tx.use_rel))
    d.nfrm0 = cfg.ds.num_sampled_frm
    d.nppf0 = int(comm["num_prop_per_frm"])
    d.nsrl = cfg.misc.srl_arg_length
    d.seq_len = cfg.ds.max_seq_length
    d.vid_w, d.vid_h = float(cfg.ds.resized_width), float(cfg.ds.resized_height)
    d.tx_dtype = L.DTYPE[tx]
    d.enc_dtype = L.VOG_F16
    return d


# ---- how sharp can this checkpoint's attention get (round 5) -----------------------------------------------
# logits = x Wq_h^T Wk_h x'^T / sqrt(d) (code/transformer_code.py:141-155, scale = sqrt(d_model)). For inputs with mean square
# r2 per feature their standard deviation is ~ ||Wq_h^T Wk_h||_F / sqrt(d) * r2. r2 of a layer's input: LayerNorm output
# (mean(g^2) + mean(b^2)) for every layer behind a LayerNorm - mul_tx layer 0 reads obj_tx's output -; 0.25 for obj_tx layer 0,
# which reads the ReLU'd encoder outputs (measured 0.13-0.3 on synthetic and heavy-tailed features). Softmax turns an ABSOLUTE
# logit error into a RELATIVE probability error and 16-bit operands make a logit error proportional to this scale, so the
# statistic orders checkpoints by how much operand precision they need. Envelope (oracle rounding model + GPU goldens
# full/cfg2_sharp{8,10,12,16}): bf16 <= ~4 (x 4: 3.2; x 8 = 12.6 measured 2.05e-3), f16 <= 20 (x 8: 3.6e-4,
# x 10 = 19.8; x 12 = 28.5 measured 9.0e-4 - inside the bound but without margin), beyond: fp32.
F16_SHARPNESS_MAX = 20.0
BF16_SHARPNESS_MAX = 4.0
# Stacks of 2-3 encoder layers (EXPTS.md:186-189): every layer behind the first reads LayerNorm outputs (r2 ~ 1, where obj_tx
# layer 0 reads r2 ~ 0.25) and a sharp layer's output error feeds the next sharp layer's logits. Rounding model, f16, worst of two
# seeds: 2 layers x 5 / 6 / 7 -> 3.9e-4 / 7.9e-4 / 1.1e-3; 3 layers x 5 / 6 / 7 / 8 -> 5.1e-4 / 1.0e-3 / 1.7e-3 / 3.8e-3
# (sharpness 4.9 / 7.1 / 9.6 / 12.6). GPU goldens: full/vog_spat_3layers_sharp4 (3.1: f16), full/vog_spat_3layers_sharp8 (fp32 path).
F16_SHARPNESS_MAX_DEEP = 5.0
# Round 6: hi + lo f16 operands (`tx_split`): everything that feeds attention logits - the encoders, the QKV projections,
# Q.K^T, and the tails whose output is another layer's input - carries a second 16-bit operand t16(x - t16(x)) and runs three
# MFMAs per product (~2^-21 relative operand error); P.V, the last tail, the BiLSTM and the heads stay f16. Rounding model
# (scratch/r6_quant_split*.py) at wq / wk x 16 / 24 / 32 / 48 (sharpness 51 / 114 / 202 / 455): 2.6e-4 / 3.4e-4 / 4.4e-4 /
# 1.1e-3; measured on the GPU against the reference goldens: x 12 1.5e-4 (plain f16: 9.0e-4), x 16 2.1e-4 (1.4e-3), x 24 6.4e-4,
# x 32 3.8e-4; cfg 3 / cfg 5 / sep at x 16 1.4-1.7e-4. Beyond x 32: the fp32 path (full/cfg2_sharp48).
# Stacks of 2-3 layers (scratch/r6_quant_split3.py): an inner layer's P.V (f16 probabilities and values) feeds the next layer's
# sharp logits - 3 layers x 8 (sharpness 12.6): 3.0e-4 with the plan above, x 12 (28.5): 2.1e-3 (2.2e-4 only with hi + lo P / V as
# well, which the kernels do not carry): the deep envelope ends at 16.
SPLIT_SHARPNESS_MAX = 210.0
SPLIT_SHARPNESS_MAX_DEEP = 16.0
# Run-time check behind the plan (vog_batch.stats: the largest |attention logit| the forwards have seen, in nats): the statistic
# above assumes isotropic inputs; what the kernels observe does not. Thresholds = the largest logits of the sharpest goldens each
# operand precision still holds 1e-3 on (tests/test_gpu_forward.py::test_logit_scale_guard prints them).
# Measured (GPU, cfg-2 goldens; mul_tx reports the bound max|x| + max|y| of its separable logits): x 1: 0.5, x 8: 36, x 10: 63 (f16
# 5.5e-4), x 16: 145 (f16 1.4e-3, hi + lo 2.1e-4), x 32: ~580 (hi + lo 3.8e-4), x 48: ~1300 (fp32 path).
F16_LOGIT_MAX = 75.0
BF16_LOGIT_MAX = 12.0
SPLIT_LOGIT_MAX = 700.0


def f16_sharpness_max(obj_layers: int, mul_layers: int) -> float:
    return F16_SHARPNESS_MAX if max(int(obj_layers), int(mul_layers)) <= 1 else F16_SHARPNESS_MAX_DEEP


def split_sharpness_max(obj_layers: int, mul_layers: int) -> float:
    return SPLIT_SHARPNESS_MAX if max(int(obj_layers), int(mul_layers)) <= 1 else SPLIT_SHARPNESS_MAX_DEEP


def attention_sharpness(sd, n_heads_obj: int, n_heads_mul: int) -> float:
    """max over the encoder layers of obj_txf / mult_txf of ||Wq_h^T Wk_h||_F / sqrt(d) * r2(layer input). sd: numpy / torch
    state dict under the reference's key names."""
    def arr(k):
        v = sd[k]
        return (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(np.float64)

    keys = {(k[7:] if k.startswith("module.") else k): k for k in sd}

    def ln_r2(stack, layer):
        for g, b in (("weight", "bias"), ("gamma", "beta")):
            kg = f"{stack}.encoder.layers.{layer}.feedforward.layernorm.{g}"
            if kg in keys:
                return float((arr(keys[kg]) ** 2).mean() + (arr(keys[kg.replace(g, b)]) ** 2).mean())
        return 1.0

    n_obj = 1 + max([int(k.split(".")[3]) for k in keys if k.startswith("obj_txf.encoder.layers.")], default=-1)
    worst = 0.0
    for k in keys:
        if not k.endswith("selfattn.layer.wq.weight"):
            continue
        stack, layer = k.split(".")[0], int(k.split(".")[3])
        wq, wk = arr(keys[k]), arr(keys[k.replace("wq.weight", "wk.weight")])
        d = wq.shape[0]
        if layer > 0:
            r2 = ln_r2(stack, layer - 1)
        elif stack == "mult_txf" and n_obj > 0:
            r2 = ln_r2("obj_txf", n_obj - 1)
        else:
            r2 = 0.25
        H = n_heads_obj if stack == "obj_txf" else n_heads_mul
        bounds = np.cumsum([0] + [len(c) for c in np.array_split(np.arange(d), H)])      # torch.chunk sizes (171/171/170)
        for h in range(H):
            a, b = wq[bounds[h]:bounds[h + 1]], wk[bounds[h]:bounds[h + 1]]
            worst = max(worst, float(np.linalg.norm(a.T @ b)) / float(np.sqrt(d)) * r2)
    return worst


# ---- persistent BiLSTM: how many forwards may be in flight ---------------------------------------------
# The layer kernel needs its 64 workgroups (one CU each) resident at the same time; they wait for each
# other's hidden states. Four such kernels fill the chip exactly; a fifth one (or another long-lived
# resident kernel, e.g. an RCCL collective waiting for its peers) can leave every one of them short of CUs
# until the hand-off times out and poisons the output with NaN. The engine therefore keeps a per-device
# BOOK: every forward it issues (slot / group launch, eager call) belongs to one of N lanes - a stream
# gets the next lane the first time it is seen - and a lane remembers the stream that used it last: a
# launch from another stream first waits for that stream (`wait_stream`), so the forwards of a lane are
# serialised on the GPU whatever the caller does and at most N layer kernels are ever resident. N = 4; while
# a cross-rank gather is pending the last lane yields to it (`_max_inflight`, `dist.pending_collective`).
# Streams that keep to themselves (bench.py: one slot per stream; the evaluator: one stream)
# never wait.
_LANE_BOOK: Dict[int, Dict[int, "torch.cuda.Stream"]] = {}
_STREAM_LANE: Dict[int, Dict[int, int]] = {}


def _dev_index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def _multi_rank_policy() -> bool:
    """A process group with more than one rank exists (or VOG_FORCE_MULTI_RANK_LANES=1: the same policy with one rank, for
    measuring what it costs - bench.py with VOG_BENCH_FORCE_DIST=1)."""
    if os.environ.get("VOG_FORCE_MULTI_RANK_LANES") == "1":
        return True
    try:
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    except Exception:
        return False


def _max_inflight() -> int:
    """Forwards in flight per device. Always 4 (round 3: 3 with a multi-rank group, permanently - 10 % of the throughput for a
    collective that runs once per 32 batches). The CUs an RCCL kernel needs are now reserved only WHILE a collective is
    pending: `dist.RecordRing` posts an event behind every gather and the LAST lane's next forward waits for it
    (`_lane_enter`), the gather itself starts behind every forward issued before it - so at most 3 forwards ever share the
    chip with a collective, and between collectives all 4 lanes run."""
    return 4


def _lane_enter(device: torch.device, stream: Optional["torch.cuda.Stream"]) -> int:
    """Call before enqueuing a forward on `stream` (None = the current stream). Returns its lane."""
    st = stream if stream is not None else torch.cuda.current_stream(device)
    i = _dev_index(device)
    m = _STREAM_LANE.setdefault(i, {})
    key = int(st.cuda_stream)
    if key not in m:
        m[key] = len(m)                     # first come, first served; later streams share round-robin
    lane = m[key] % _max_inflight()
    book = _LANE_BOOK.setdefault(i, {})
    if lane == _max_inflight() - 1 and _multi_rank_policy():
        from . import dist as D                     # the last lane yields to a pending cross-rank gather
        gate = D.pending_collective(i)
        if gate is not None:
            st.wait_event(gate)
    prev = book.get(lane)
    if prev is not None and prev.cuda_stream != st.cuda_stream:
        st.wait_stream(prev)
    book[lane] = st
    return lane


class VogEngine:
    def __init__(self, cfg, comm, device: Optional[torch.device] = None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.VogError("VogEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                             "there is no CPU fallback on the product path")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.cfg = cfg
        self.desc = model_desc_from_cfg(cfg, comm)
        self.conc_type = cfg.ds.conc_type
        self.sep = self.conc_type in ("sep", "svsq")
        h = C.c_void_p()
        L.check(self.lib.vog_ctx_create(C.byref(self.desc), C.byref(h)), "vog_ctx_create")
        self.ctx = h
        self._ws: Dict[Tuple[int, int, int, int], torch.Tensor] = {}
        self.weights_epoch = 0          # bumped by load_state_dict: slots / groups captured before it refuse to launch
        self._graphs: Dict[tuple, C.c_void_p] = {}
        self._finalized = False
        self.use_graph = bool(cfg.hip.use_graph) if "hip" in cfg else True
        hip = cfg.get("hip", {}) if hasattr(cfg, "get") else {}
        self.tx_request = hip.get("tx_dtype", "auto") if hasattr(hip, "get") else "auto"
        # stalled BiLSTM hand-offs (round 5): the persistent layer kernel counts a timed-out launch into a word of PINNED host
        # memory (vog_batch.fault); the host reads it without synchronising the device wherever it is about to hand out or
        # reuse results (`check`, `Slot.check`): a stall is a VogError at the API, never NaN scores with rc 0. Word 0: the
        # eager forwards of this engine; slots own theirs.
        self._fault = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._fault_seen = 0
        self.stalls = 0                 # stalled forwards seen so far (all slots)
        self.sharpness = 0.0            # attention_sharpness of the loaded checkpoint
        self.precise = None             # precise.PreciseForward when this checkpoint runs the fp32 path
        self.plan = "f16"               # operand precision in use: bf16 / f16 / split (hi + lo f16) / f32
        # largest |attention logit| (obj_tx, mul_tx) the forwards have seen: pinned host words the prediction head folds its
        # forward's maxima into (vog_batch.stats) - `observed_logit_max`, `check_logit_scale`
        self._stats = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._scale_warned = False

    # ---- weights -------------------------------------------------------------
    def expected_weights(self) -> Dict[str, int]:
        n = self.lib.vog_ctx_num_weights(self.ctx)
        return {self.lib.vog_ctx_weight_name(self.ctx, i).decode():
                int(self.lib.vog_ctx_weight_numel(self.ctx, i)) for i in range(n)}

    def load_state_dict(self, sd) -> None:
        """Register weights (torch tensors or numpy arrays) under the reference's
        key names (utils/trn_utils.py:534-593 semantics: `module.` prefix and
        legacy LayerNorm gamma/beta names are accepted by the library)."""
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            L.check(self.lib.vog_ctx_set_weight(self.ctx, k.encode(), a.ctypes.data, a.size),
                    f"vog_ctx_set_weight({k})")
        self._drop_graphs()
        self._ws.clear()                        # (the workspace plan depends on the precision plan decided below)
        self._sd_ref = sd                       # (kept for a plan raised at run time: `check_logit_scale`)
        # precision plan of THIS checkpoint (round 5 / 6), decided from the weights before they are converted: `auto` = f16
        # operands inside their envelope, hi + lo f16 operands (`tx_split`: three MFMAs for everything that feeds attention logits)
        # up to ~12 x that sharpness, the fp32 path beyond; an explicit bf16 / f16 request is honoured but the envelope is reported
        has_tx = self.cfg.mdl.name in ("vgrnd", "vog")
        self.sharpness = attention_sharpness(sd, int(self.desc.obj_heads), int(self.desc.mul_heads)) if has_tx else 0.0
        nl = (self.desc.obj_layers if has_tx else 0, self.desc.mul_layers if self.cfg.mdl.name == "vog" else 0)
        f16_max, split_max = f16_sharpness_max(*nl), split_sharpness_max(*nl)
        split_ok = bool(self.lib.vog_ctx_split_supported(self.ctx, 1 if self.conc_type == "svsq" else 4))
        want_split = self.tx_request == "split" or (self.tx_request == "auto" and f16_max < self.sharpness <= split_max and split_ok)
        if self.tx_request == "split" and not split_ok:
            raise L.VogError("cfg.hip.tx_dtype = split: this model shape has no hi + lo kernels (gt5-sized sequences, fused encoders "
                             "and tails: vog_ctx_split_supported); use auto / f32")
        want_f32 = self.tx_request == "f32" or (self.tx_request == "auto" and self.sharpness > f16_max and not want_split)
        self.set_option("tx_split", int(want_split))
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()            # nothing in flight may still read the buffers finalize frees
            L.check(self.lib.vog_ctx_finalize(self.ctx), "vog_ctx_finalize")
        self._finalized = True
        self.weights_epoch += 1
        self.precise = None
        self.plan = "split" if want_split else ("f32" if want_f32 else ("bf16" if self.desc.tx_dtype == L.VOG_BF16 else "f16"))
        self._stats.zero_()
        self._scale_warned = False
        if want_f32:
            from .precise import PreciseForward
            self.precise = PreciseForward(self, sd)
        elif not want_split and self.sharpness > (min(BF16_SHARPNESS_MAX, f16_max) if self.tx_request == "bf16" else f16_max):
            import warnings
            warnings.warn(f"attention sharpness {self.sharpness:.1f} of this checkpoint is outside the envelope in which "
                          f"tx_dtype={self.tx_request} holds 1e-3 on pred_scores (DESIGN.md section 2); use tx_dtype=auto")

    # ---- observed logit scale (round 6) ----------------------------------------
    def observed_logit_max(self):
        """(obj_tx, mul_tx): the largest |attention logit| (nats, after bias and 1 / sqrt(d)) any forward of this engine has
        computed since the weights were loaded - written by the kernels (vog_attn_args.logit_max), folded into pinned host memory
        by the prediction head; a host read, no device synchronisation (forwards still in flight are not in it yet). 0.0 for a
        stack whose kernels do not report (the long-sequence kernels of p100)."""
        w = self._stats[:2].numpy().view(np.float32)
        return float(w[0]), float(w[1])

    def check_logit_scale(self, escalate: bool = True) -> bool:
        """The run-time side of the precision plan: the per-checkpoint statistic (`attention_sharpness`) assumes isotropic inputs;
        this compares what the attention kernels actually saw with the logit scale the operand precision in use holds 1e-3 on. Past
        it: a warning (the batches already returned may be off by more than 1e-3 in pred_scores), and with `escalate` the plan is
        raised for everything issued from now on (f16 -> hi + lo -> fp32; slots captured before refuse to launch, like after any
        `load_state_dict`). Returns True while the plan holds. Called by `mdl_base.forward` and the evaluator after their
        synchronisations; cheap (two host words)."""
        if self.plan == "f32" or self.tx_request not in ("auto", "bf16", "f16", "split"):
            return True
        lim = {"bf16": BF16_LOGIT_MAX, "f16": F16_LOGIT_MAX, "split": SPLIT_LOGIT_MAX}[self.plan]
        seen = max(self.observed_logit_max())
        if seen <= lim:
            return True
        import warnings
        nxt = "split" if (self.plan in ("bf16", "f16") and seen <= SPLIT_LOGIT_MAX and
                          bool(self.lib.vog_ctx_split_supported(self.ctx, 1 if self.conc_type == "svsq" else 4))) else "f32"
        if not self._scale_warned:
            warnings.warn(f"attention logits of up to {seen:.0f} nats observed: outside the range in which {self.plan} operands hold "
                          f"1e-3 on pred_scores ({lim:.0f}); results so far may exceed the bound" +
                          (f" - re-planning to '{nxt}' for the forwards issued from now on" if escalate and self.tx_request == "auto" else ""))
            self._scale_warned = True
        if escalate and self.tx_request == "auto":
            req, self.tx_request = self.tx_request, nxt
            try:
                self.load_state_dict(self._sd_ref)
            finally:
                self.tx_request = req
        return False

    # ---- stalled hand-offs ---------------------------------------------------
    def _stalled(self, n: int, where: str):
        """n forwards timed out: degrade to the step-launch BiLSTM for everything issued from now on (eager forwards and new
        slots; it needs no co-residency and cannot stall) and raise."""
        self.stalls += n
        try:
            self.set_option("lstm_persistent", 0)
        except Exception:
            pass
        raise L.VogError(
            f"{n} BiLSTM layer launch(es) of {where} lost their hand-off (the persistent layer kernel's 64 workgroups were not "
            "co-resident within ~1 s: more than 4 forwards in flight on this device, or another resident kernel holding CUs); "
            "their outputs are NaN and must be discarded. The engine now uses the step-launch BiLSTM (lstm_persistent = 0) "
            "for eager forwards and new slots: re-run the affected batches. (reference: utils/mdl_srl_utils.py:114-169 "
            "cannot fail by scheduling, so this is an error, not a result)")

    def check(self) -> None:
        """Raise VogError if an eager forward issued by this engine stalled since the last check. A host read of pinned
        memory: no device synchronisation - a forward still in flight is judged by the next call. Called by `forward` (for
        the forwards before it), `mdl_base.forward`, and the evaluator wherever it has synchronised."""
        n = int(self._fault[0])
        if n != self._fault_seen:
            k, self._fault_seen = n - self._fault_seen, n
            self._stalled(k, "this engine's eager path")

    # ---- workspace -----------------------------------------------------------
    def workspace(self, B: int, ncmp: int, T: int) -> torch.Tensor:
        # one workspace per (shape, stream): two forwards of one shape issued on different streams must
        # not share LSTM hand-off words / q,k,v fragments / the implicit intermediates
        key = (B, ncmp, T, int(L.stream_ptr()))
        ws = self._ws.get(key)
        if ws is None:
            n = self.lib.vog_workspace_bytes(self.ctx, B, ncmp, T)
            if n < 0:
                raise L.VogError("vog_workspace_bytes failed (weights not finalized?)")
            ws = torch.empty(int(n), dtype=torch.uint8, device=self.device)
            L.check(self.lib.vog_workspace_init(self.ctx, B, ncmp, T, ws.data_ptr(), ws.numel(),
                                                L.stream_ptr()), "vog_workspace_init")
            self._ws[key] = ws
        return ws

    def stage(self, B, ncmp, T, name, dtype, shape) -> torch.Tensor:
        """View of a named intermediate inside the workspace (parity tests)."""
        off, nb = C.c_int64(), C.c_int64()
        L.check(self.lib.vog_workspace_stage(self.ctx, B, ncmp, T, name.encode(), C.byref(off),
                                             C.byref(nb)), f"stage {name}")
        ws = self.workspace(B, ncmp, T)
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        assert n <= nb.value, (name, n, nb.value)
        return ws[off.value: off.value + n].view(dtype).view(*shape)

    # ---- forward -------------------------------------------------------------
    def _geometry(self, inp):
        B = inp["srl_arg_words_ind"].shape[0]
        ncmp = inp["new_srl_idxs"].shape[1] if "new_srl_idxs" in inp else inp["num_cmp_msk"].shape[1]
        d = self.desc
        nc_v = ncmp if self.sep else 1
        NP = ncmp * d.nfrm0 * d.nppf0 if not self.sep else d.nfrm0 * d.nppf0
        return B, ncmp, nc_v, NP

    def make_batch(self, inp: Dict[str, torch.Tensor], T: Optional[int] = None,
                   with_pred: bool = True, pred_rec: Optional[torch.Tensor] = None):
        """Validate a batch dict, allocate outputs, fill the C struct. `pred_rec`: caller-owned
        storage for the packed prediction records ([B, record_words] fp32, contiguous), e.g. a
        slice of one buffer shared by the slots in flight so that ONE all-gather exchanges them."""
        d = self.desc
        B, ncmp, nc_v, NP = self._geometry(inp)
        for k in NSRL_KEYS_I64:
            assert inp[k].dtype == torch.int64 and inp[k].is_cuda, k
        for k in F32_KEYS:
            assert inp[k].dtype == torch.float32 and inp[k].is_cuda, k
        nv = inp["srl_arg_words_ind"].shape[1]
        assert nv == (ncmp if self.sep else 1), "language axis does not match conc_type"
        assert inp["srl_arg_words_ind"].shape[2:] == (d.nsrl, d.seq_len)
        assert inp["srl_tag_word_ind"].shape == inp["srl_arg_word_mask"].shape \
            if "srl_tag_word_ind" in inp else True
        vis_lead = (B, ncmp) if self.sep else (B,)
        assert tuple(inp["pad_region_feature"].shape) == vis_lead + (NP, d.prop_dim)
        assert tuple(inp["pad_proposals"].shape) == vis_lead + (NP, 7)
        assert tuple(inp["seg_feature_for_frms"].shape) == vis_lead + (NP // d.nppf0, d.seg_dim)
        if T is None:
            # the reference does the same host read (mdl_vog.py:257 `.max().item()`)
            T = int(inp["srl_arg_word_mask_len"].max().item())
        dev = self.device
        out = {
            "mdl_outs": torch.empty(B, nc_v, d.nsrl, NP, device=dev),
            "mdl_outs_eval": torch.empty(B, nc_v, d.nsrl, NP, device=dev),
        }
        if self.sep:
            out["vidf_outs"] = torch.empty(B, ncmp, device=dev)
            out["fin_scores_loss"] = torch.empty(B, ncmp, d.nsrl, device=dev)
            out["fin_scores"] = torch.empty(B, ncmp, device=dev)
        rec = None
        if with_pred:
            rb = int(self.lib.vog_pred_record_bytes(ncmp, d.nsrl, d.nfrm0))
            if pred_rec is not None:
                assert pred_rec.is_cuda and pred_rec.dtype == torch.float32 and pred_rec.is_contiguous() \
                    and tuple(pred_rec.shape) == (B, rb // 4), "pred_rec storage has the wrong shape"
                rec = pred_rec
            else:
                rec = torch.empty(B, rb // 4, dtype=torch.float32, device=dev)
            out["pred_rec"] = rec
        b = L.Batch()
        b.B, b.ncmp, b.T = B, ncmp, T
        for k in NSRL_KEYS_I64 + F32_KEYS:
            setattr(b, k, L.ptr(inp[k]))
        if self.sep:
            v = inp["verb_ind_in_srl"]
            if v.shape[1] == 1 and ncmp > 1:
                v = v.expand(-1, ncmp).contiguous()
            out["_verb"] = v
            b.verb_ind_in_srl = L.ptr(v)
            b.vidf_outs = L.ptr(out["vidf_outs"])
            b.fin_scores_loss = L.ptr(out["fin_scores_loss"])
            b.fin_scores = L.ptr(out["fin_scores"])
        b.mdl_outs = L.ptr(out["mdl_outs"])
        b.mdl_outs_eval = L.ptr(out["mdl_outs_eval"])
        b.pred_rec = L.ptr(rec)
        b.fault = self._fault.data_ptr()            # (slots replace it with their own word before they capture)
        b.stats = self._stats.data_ptr() if with_pred else None
        return b, out, (B, ncmp, T)

    def forward(self, inp: Dict[str, torch.Tensor], T: Optional[int] = None,
                with_pred: bool = True) -> Dict[str, torch.Tensor]:
        """Eager launch sequence on the current stream (fresh output tensors)."""
        assert self._finalized, "load_state_dict first"
        self.check()
        self.check_logit_scale()                # (two host words; raises the plan if the forwards so far saw sharper logits than planned)
        with torch.cuda.device(self.device):
            b, out, (B, ncmp, T) = self.make_batch(inp, T, with_pred)
            _lane_enter(self.device, None)
            ws = self.workspace(B, ncmp, T)
            if self.precise is not None:                    # checkpoints outside the f16 envelope: the fp32 path alone (precise.py)
                self.precise.run(inp, out, T=T)
            else:
                L.check(self.lib.vog_forward(self.ctx, C.byref(b), ws.data_ptr(), ws.numel(),
                                             L.stream_ptr()), "vog_forward")
        out["_keepalive"] = (inp, ws)
        return out

    # ---- persistent slots (graph replay; what bench.py and the evaluator use) ----
    def make_slot(self, inp: Dict[str, torch.Tensor], T: Optional[int] = None,
                  with_pred: bool = True, graph: Optional[bool] = None,
                  pred_rec: Optional[torch.Tensor] = None, share_ws_with: Optional["Slot"] = None) -> "Slot":
        """`share_ws_with`: reuse another slot's workspace (same shapes). Only for slots that are launched on the SAME
        stream one after the other (every forward re-initialises the state it needs in its prologue): N input sets
        cycling through a few workspaces, the layout of a serving loop whose requests arrive in fresh buffers."""
        return Slot(self, inp, T, with_pred, self.use_graph if graph is None else graph, pred_rec, share_ws_with)

    def record_words(self, ncmp: int) -> int:
        """fp32 words of one packed prediction record (boxes || scores || pred_cmp)."""
        return int(self.lib.vog_pred_record_bytes(ncmp, self.desc.nsrl, self.desc.nfrm0)) // 4

    def make_batched(self, inps, with_pred: bool = True, graph: bool = True, pred_rec=None) -> "Batched":
        """Several requests (input dicts of the same shape) served as ONE forward: their rows live back to back in one slot
        (dynamic batching; rows never interact, every member gets the outputs of its own forward). At cfg 2 four bs=4 requests
        per forward run at 77 k queries/s against 56 k for four separate forwards in flight: every kernel of the chain is
        four times wider, the BiLSTM uses 16 of its 16 MFMA columns, and a batch costs a quarter of the launches."""
        return Batched(self, inps, with_pred, graph, pred_rec)

    def make_group(self, inps, with_pred: bool = True, graph: bool = True, pred_rec=None) -> "Group":
        """Several batches whose LANGUAGE encoder runs once for all of them (W_hh is streamed once
        per recurrent step instead of once per batch per step; include/vog_hip.h, "language
        encoder over a group"). Every member keeps its own inputs, outputs and workspace, and its
        outputs equal its stand-alone forward up to fp32 summation order."""
        return Group(self, inps, with_pred, graph, pred_rec)

    def set_option(self, name: str, value: int) -> None:
        """Integer options of the context: 'lstm_persistent', 'fused_tail', 'pair_launches', ... (include/vog_hip.h)."""
        L.check(self.lib.vog_ctx_set_int(self.ctx, name.encode(), int(value)), f"vog_ctx_set_int({name})")


    def time_kernel(self, slot: "Slot", name: str, iters: int = 50) -> float:
        us = C.c_float()
        L.check(self.lib.vog_time_kernel(self.ctx, C.byref(slot.batch), slot.ws.data_ptr(),
                                         slot.ws.numel(), name.encode(), iters, L.stream_ptr(),
                                         C.byref(us)), f"vog_time_kernel({name})")
        return float(us.value)

    def _drop_graphs(self):
        for g in self._graphs.values():
            self.lib.vog_graph_destroy(g)
        self._graphs.clear()

    def unpack_pred(self, rec: torch.Tensor, ncmp: int):
        """Packed records -> the reference's {'boxes','scores','indexs'} tensors
        (eval_vsrl_corr.py:216-220). TEMP returns float zeros for indexs (:338-340)."""
        d = self.desc
        B = rec.shape[0]
        nb = d.nsrl * ncmp * d.nfrm0
        boxes = rec[:, : nb * 7].reshape(B, d.nsrl, ncmp, d.nfrm0, 7)
        scores = rec[:, nb * 7: nb * 8].reshape(B, d.nsrl, ncmp, d.nfrm0)
        idx = rec[:, nb * 8:].contiguous().view(torch.int64).reshape(B, d.nsrl, d.nfrm0)
        if self.conc_type == "temp":
            idx = torch.zeros(B, d.nsrl, d.nfrm0, dtype=torch.float32, device=rec.device)
        return {"boxes": boxes, "scores": scores, "indexs": idx}

    def __del__(self):
        try:
            self._drop_graphs()
            if getattr(self, "ctx", None):
                self.lib.vog_ctx_destroy(self.ctx)
        except Exception:
            pass


class Slot:
    """Persistent device buffers for one batch shape + a captured hipGraph.

    Inputs live at fixed addresses (H2D copies land here directly), so one
    forward is a single hipGraphLaunch of ~50 kernel nodes."""

    def __init__(self, eng: VogEngine, inp, T, with_pred, graph, pred_rec=None, share_ws_with=None):
        self.eng = eng
        self.epoch = eng.weights_epoch
        # the slot OWNS its input buffers (update_inputs / the device batch assembly write into them): a
        # tensor that already lives on the device is copied, never aliased
        def own(v):
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
            d = t.to(eng.device).contiguous()
            return d.clone() if d.data_ptr() == t.data_ptr() else d
        self.inp = {k: own(v) for k, v in inp.items()}
        with torch.cuda.device(eng.device):
            self.batch, self.out, (self.B, self.ncmp, self.T) = eng.make_batch(self.inp, T, with_pred, pred_rec)
            self._fault = torch.zeros(16, dtype=torch.int32).pin_memory()      # this slot's stall counter (see VogEngine.check)
            self._fault_seen = 0
            self.batch.fault = self._fault.data_ptr()
            n = eng.lib.vog_workspace_bytes(eng.ctx, self.B, self.ncmp, self.T)
            if share_ws_with is not None:
                o = share_ws_with
                assert (o.B, o.ncmp, o.T) == (self.B, self.ncmp, self.T) and o.ws.numel() == int(n), \
                    "a shared workspace needs slots of the same shape"
                self.ws = o.ws
            else:
                self.ws = torch.empty(int(n), dtype=torch.uint8, device=eng.device)
                L.check(eng.lib.vog_workspace_init(eng.ctx, self.B, self.ncmp, self.T, self.ws.data_ptr(),
                                                   self.ws.numel(), L.stream_ptr()), "vog_workspace_init")
            self.graph = None
            if graph:
                torch.cuda.synchronize()
                cap = torch.cuda.Stream(device=eng.device)
                g = C.c_void_p()
                L.check(eng.lib.vog_graph_capture(eng.ctx, C.byref(self.batch), self.ws.data_ptr(),
                                                  self.ws.numel(), cap.cuda_stream, C.byref(g)),
                        "vog_graph_capture")
                self.graph = g
                torch.cuda.synchronize()

    def feed_from(self, staging, assembler=None, via: str = "zero_copy", part: Optional[int] = None) -> "Slot":
        """Make this slot HOST-FED: re-capture its graph with the batch's way onto the device in front of the forward
        (vog_graph_capture_fed), reading `staging.host` - a `dat_loader_simple.PackedStaging`'s pinned host views - at fixed
        addresses over the host link. A step of the loop is then ONE `launch()`: the loader writes batch i + 1 into
        `staging.host[k]` (after the launch that read batch i has completed: `consumed()`), nothing is copied, staged or
        assembled by separate calls. With `assembler` (a `DeviceBatchAssembler`, SPAT / TEMP) the per-video items
        (`FWD_KEYS`) go through vog_assemble_batch into the slot's concatenated input buffers; every other key that the slot
        and the staging buffer share (same byte size) is copied by vog_copy_segments.
        `via`: "zero_copy" (above); "dma_node": the graph starts with ONE host -> device memcpy node of the packed buffer into
        `staging.dbuf` and the kernels read the device views (measured no faster than zero copy: 42 GB/s at p100, the runtime
        executes the node with a copy kernel); "device": the graph reads the device views and the CALLER moves the packed
        buffer with the copy engine on a copy stream (`staging.upload_on(copy_stream, consumer=stream)`, `launch(stream)`,
        `staging.release(stream)`; the staging buffer must have ONE device buffer, n_dev = 1): 55 GB/s on large batches and
        the copy of a slot's next batch overlaps the other slots' forwards.
        `part`: the staging buffer holds SEVERAL batches (every tensor with a leading axis over them) and this slot reads
        batch `part`: one transfer then feeds a group of slots - a transfer costs ~80 us whatever its size (8.5 MB cfg-2
        batches one per transfer: 36 GB/s; four per transfer: the link's 50+).
        (reference: the `.to(device)` of every batch tensor, code/utils/trn_utils.py:478 / :562, and the SPAT / TEMP
        concatenation of the collate step, code/dat_loader_simple.py:380-520)"""
        assert self.graph is not None, "feed_from needs a graph slot"
        eng = self.eng
        asm_keys = ()
        a = None
        assert via in ("zero_copy", "dma_node", "device")
        assert via == "zero_copy" or len(staging.dbufs) == 1, "the graph reads ONE device buffer: PackedStaging(n_dev=1)"
        src = staging.host if via == "zero_copy" else staging.dev
        if part is not None:
            src = {k: v[part] for k, v in src.items()}
        dseg = None
        if via == "dma_node":
            dseg = L.CopySeg()
            dseg.src, dseg.dst, dseg.bytes = staging.hbuf.data_ptr(), staging.dbuf.data_ptr(), staging.nbytes
        if assembler is not None:
            from .dat_loader_simple import FWD_KEYS
            asm_keys = FWD_KEYS
            a, _ = assembler.args({k: src[k] for k in FWD_KEYS}, out={k: self.inp[k] for k in FWD_KEYS},
                                  with_loss_keys=False)
        segs = []
        for k, h in src.items():
            if k in asm_keys or k not in self.inp:
                continue
            d = self.inp[k]
            nb = h.numel() * h.element_size()
            assert d.numel() * d.element_size() == nb and d.dtype == h.dtype, f"staging['{k}'] does not match the slot's input"
            segs.append((h.data_ptr(), d.data_ptr(), nb))
        assert len(segs) <= L.MAX_COPY_SEGS
        arr = (L.CopySeg * max(1, len(segs)))()
        for i, (sp_, dp_, nb) in enumerate(segs):
            arr[i].src, arr[i].dst, arr[i].bytes = sp_, dp_, nb
        with torch.cuda.device(eng.device):
            torch.cuda.synchronize()
            cap = torch.cuda.Stream(device=eng.device)
            g = C.c_void_p()
            L.check(eng.lib.vog_graph_capture_fed(eng.ctx, C.byref(self.batch), self.ws.data_ptr(), self.ws.numel(),
                                                  C.byref(dseg) if dseg is not None else None,
                                                  C.byref(a) if a is not None else None, arr, len(segs),
                                                  cap.cuda_stream, C.byref(g)), "vog_graph_capture_fed")
            eng.lib.vog_graph_destroy(self.graph)
            self.graph = g
            torch.cuda.synchronize()
        self.feed = staging
        self.fed_keys = tuple(asm_keys) + tuple(k for k in src if k in self.inp and k not in asm_keys)
        self._consumed = None
        return self

    def consumed(self, stream: Optional[torch.cuda.Stream] = None) -> "torch.cuda.Event":
        """Fed slots: an event that fires when everything launched on `stream` so far (hence the last `launch()` there, and
        its reads of the staging buffer) has completed; the loader waits for it before it writes the next batch."""
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.eng.device))
        return ev

    def update_inputs(self, inp, check_lengths: bool = True):
        """Copy a new batch (same shapes, sentence lengths <= the T this slot was captured with) into
        the slot's buffers. T is baked into the captured graph and the workspace: a
        longer sentence would index the LSTM schedule out of its rows, so it is refused here (capture
        the slot with T = cfg.ds.max_seq_length when the lengths are not known in advance)."""
        lens = inp.get("srl_arg_word_mask_len") if check_lengths else None      # (the check reads the lengths: a host sync)
        if lens is not None:
            mx = int((lens if isinstance(lens, torch.Tensor) else torch.as_tensor(lens)).max())
            if mx > self.T:
                raise ValueError(f"batch has a sentence of {mx} tokens but this slot was captured with T = {self.T}")
        for k, v in inp.items():
            if k in self.inp:
                self.inp[k].copy_(v if isinstance(v, torch.Tensor) else torch.from_numpy(v),
                                  non_blocking=True)

    def _check_epoch(self):
        if self.epoch != self.eng.weights_epoch:
            raise L.VogError("the engine's weights were re-finalized after this slot was captured: its graph "
                             "points at freed weight buffers - create a new slot")

    def check(self) -> None:
        """Raise VogError if a launch of this slot stalled since the last check (host read of pinned memory, no device
        synchronisation: call it after synchronising to judge the launches before that point)."""
        n = int(self._fault[0])
        if n != self._fault_seen:
            k, self._fault_seen = n - self._fault_seen, n
            self.eng._stalled(k, f"a slot (B = {self.B}, T = {self.T})")

    def launch(self, stream: Optional[torch.cuda.Stream] = None):
        self._launches = getattr(self, "_launches", 0) + 1
        if (self._launches & 15) == 1:        # every 16th launch: the observed logit scale against the plan (may re-plan: this slot
            self.eng.check_logit_scale()      # then refuses to launch, like after any load_state_dict)
        self._check_epoch()
        self.check()                          # the launches before this one
        _lane_enter(self.eng.device, stream)
        sp = L.stream_ptr(stream)
        # (fp32 path: a fed slot's graph still has to run - it moves / assembles the batch - anything else is skipped)
        if self.eng.precise is None or getattr(self, "feed", None) is not None:
            if self.graph is not None:
                L.check(self.eng.lib.vog_graph_launch(self.graph, sp), "vog_graph_launch")
            else:
                L.check(self.eng.lib.vog_forward(self.eng.ctx, C.byref(self.batch), self.ws.data_ptr(),
                                                 self.ws.numel(), sp), "vog_forward")
        self._precise(stream)
        return self.out

    def _precise(self, stream):
        """Checkpoints outside the f16 envelope: the fp32 forward runs behind whatever the slot's launch did (copies /
        assembly of a fed slot included) on the same stream and overwrites the slot's outputs."""
        pf = self.eng.precise
        if pf is None:
            return
        with torch.cuda.device(self.eng.device):
            if stream is None:
                pf.run(self.inp, self.out, T=self.T)
            else:
                with torch.cuda.stream(stream):
                    pf.run(self.inp, self.out, T=self.T)

    def __del__(self):
        try:
            if self.graph is not None:
                self.eng.lib.vog_graph_destroy(self.graph)
        except Exception:
            pass


def _shares_hw_queue(a: "torch.cuda.Stream", b: "torch.cuda.Stream", scratch: torch.Tensor, spin_cycles: int, spin_s: float) -> bool:
    """Do two HIP streams sit on the same hardware queue? A spin kernel on `a`, then a tiny kernel on `b`: on a shared queue
    the second one is dispatched behind the first (the runtime deals its streams onto 4 hardware queues)."""
    import time
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
    t0 = time.perf_counter()
    with torch.cuda.stream(b):
        scratch.add_(1)
        ev = torch.cuda.Event()
        ev.record(b)
    ev.synchronize()
    dt = time.perf_counter() - t0
    a.synchronize()
    return dt > 0.5 * spin_s


def paired_copy_streams(streams, device, candidates: int = 16):
    """One copy stream per forward stream ON THE SAME HARDWARE QUEUE (None where no candidate matched). An event recorded on a copy
    stream is a barrier packet in that stream's hardware queue until the transfer has landed; everything behind it in the queue
    waits - harmless when that is the forward that needs the transfer anyway, 30 % of the host-fed rate when it is another
    slot's forward (cfg 2: 20.7 k queries/s with aligned pairs, 14.9 k misaligned, 19.6 k with the transfers on the forward
    streams themselves; profiles/round4_host_fed.md)."""
    import time
    with torch.cuda.device(device):
        scratch = torch.zeros(64, device=device)
        cyc = 200_000
        for _ in range(2):                         # calibrate the spin to ~0.4 ms (the counter's rate differs between parts)
            torch.cuda._sleep(cyc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            torch.cuda._sleep(cyc)
            torch.cuda.synchronize()
            spin_s = time.perf_counter() - t0
            cyc = max(20_000, min(50_000_000, int(cyc * 4e-4 / max(spin_s, 1e-6))))
        t0 = time.perf_counter()
        torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        spin_s = time.perf_counter() - t0
        cands = [torch.cuda.Stream(device=device) for _ in range(candidates)]
        out, used = [], set()
        for fs in streams:
            hit = None
            for i, c in enumerate(cands):
                if i in used:
                    continue
                if _shares_hw_queue(fs, c, scratch, cyc, spin_s) and _shares_hw_queue(fs, c, scratch, cyc, spin_s):
                    hit = i
                    break
            if hit is not None:
                used.add(hit)
            out.append(cands[hit] if hit is not None else None)
    return out


class FedPipeline:
    """The host-fed serving / validation loop as an object: `streams` forward streams x `slots_per_stream` fed slots
    (`Slot.feed_from(..., via="device")`), every slot with its own packed pinned staging buffer and device copy, one copy stream
    per forward stream. Per batch:

        st = pipe.next_staging()        # pinned host views of the slot that is next in turn; its previous transfer has left it
        st.fill(raw_batch)              # (or write into st.host[k] in place)
        slot = pipe.submit()            # ONE transfer (copy engine) + ONE graph launch: assembly + word arrays + forward
        ... slot.out is valid once `pipe.done(slot)` has fired (stream order on `pipe.stream_of(slot)`)

    Two slots per stream: the transfer of a slot's next batch never waits for that slot's previous forward (a copy queue
    waiting for a compute queue cost 60 us per step at cfg 2); one copy stream per forward stream: no transfer queues behind
    another slot's. 21.6 k queries/s at cfg 2 and 24.5 k at cfg 3 from pinned host memory (link: 46 / 52 GB/s), 1.70 k at
    cfg 4 (56.6 GB/s = the link); profiles/round4_host_fed.md. The slots of a stream share one workspace.
    (reference: the loop body `batch = {k: v.to(device)}; out = mdl(batch)` of code/utils/trn_utils.py:478-485, :562-570
    behind the collate step's SPAT / TEMP concatenation, code/dat_loader_simple.py:380-520)"""

    def __init__(self, eng: VogEngine, example_inp, spec, assembler=None, streams: int = 4, slots_per_stream: int = 2,
                 T: Optional[int] = None, with_pred: bool = True, pred_rec=None, stream_pool=None, copy_streams=None):
        from .dat_loader_simple import PackedStaging
        self.eng = eng
        dev = eng.device
        self.n_streams, self.per = int(streams), max(1, int(slots_per_stream))
        n = self.n_streams * self.per
        self.streams = list(stream_pool[:self.n_streams]) if stream_pool else [torch.cuda.Stream(device=dev) for _ in range(self.n_streams)]
        # copy_streams: "own" = the transfer is issued on the slot's forward stream (no cross-stream events at all)
        #               default: a copy stream per forward stream on the same hardware queue (`paired_copy_streams`)
        self.copy_own = isinstance(copy_streams, str) and copy_streams == "own"
        if self.copy_own:
            self.copy_streams = []
        elif copy_streams:
            self.copy_streams = list(copy_streams)
        else:
            self.copy_streams = paired_copy_streams(self.streams, dev)
            if any(c is None for c in self.copy_streams):
                self.copy_own, self.copy_streams = True, []
        self.slots, self.stagings = [], []
        with torch.cuda.device(dev):
            for j in range(n):
                rec = None if pred_rec is None else pred_rec[j]
                sl = eng.make_slot(example_inp, T=T, with_pred=with_pred, graph=True, pred_rec=rec,
                                   share_ws_with=self.slots[j % self.n_streams] if j >= self.n_streams else None)
                st = PackedStaging(spec, dev, n_dev=1)
                sl.feed_from(st, assembler, via="device")
                self.slots.append(sl)
                self.stagings.append(st)
        self._done = [None] * n
        self._i = 0

    def next_staging(self):
        """The staging buffer of the slot that `submit()` will launch next. Blocks (host) until the transfer that last read its
        host side has completed - 2 x streams batches ago: never in steady state."""
        st = self.stagings[self._i % len(self.slots)]
        ev = st._ready[0]
        if ev is not None:
            ev.synchronize()
        return st

    def stream_of(self, slot: "Slot") -> "torch.cuda.Stream":
        return self.streams[self.slots.index(slot) % self.n_streams]

    def submit(self) -> "Slot":
        j = self._i % len(self.slots)
        self._i += 1
        u = j % self.n_streams
        st, sl, fs = self.stagings[j], self.slots[j], self.streams[u]
        if self.copy_own:
            with torch.cuda.stream(fs):
                st.dbuf.copy_(st.hbuf, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(fs)
            st._ready[0] = ev
            sl.launch(fs)
            ev2 = torch.cuda.Event()
            ev2.record(fs)
            self._done[j] = ev2
            return sl
        st.upload_on(self.copy_streams[u % len(self.copy_streams)], consumer=fs)
        sl.launch(fs)
        st.release(fs)
        self._done[j] = st._free[0]
        return sl

    def done(self, slot: "Slot"):
        """Event behind the slot's last forward (its outputs are complete, its device staging buffer is free)."""
        return self._done[self.slots.index(slot)]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()


LANG_KEYS = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture",
             "srl_arg_inds_msk")


class _MemberView:
    """One request of a `Batched` slot: views of its rows of the shared input / output tensors."""

    def __init__(self, big: "Slot", lo: int, hi: int):
        self.inp = {k: v[lo:hi] for k, v in big.inp.items() if isinstance(v, torch.Tensor) and v.dim() >= 1}
        self.out = {k: v[lo:hi] for k, v in big.out.items() if isinstance(v, torch.Tensor) and v.dim() >= 1}


class Batched:
    """G requests as one forward (`VogEngine.make_batched`): ONE slot whose batch axis holds the members back to back."""

    def __init__(self, eng: VogEngine, inps, with_pred=True, graph=True, pred_rec=None):
        assert len(inps) >= 1
        dev = eng.device

        def dev_t(v):
            return (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).to(dev)

        keys = list(inps[0].keys())
        sizes = [int(dev_t(i["srl_arg_words_ind"]).shape[0]) for i in inps]
        assert len(set(sizes)) == 1, "batched requests must have the same batch size"
        cat = {k: torch.cat([dev_t(i[k]) for i in inps], dim=0).contiguous() for k in keys}
        self.eng, self.B = eng, sizes[0]
        self.big = eng.make_slot(cat, with_pred=with_pred, graph=graph, pred_rec=pred_rec)
        self.slots = [_MemberView(self.big, m * self.B, (m + 1) * self.B) for m in range(len(inps))]
        self.graph = self.big.graph

    @property
    def out(self):
        return [s.out for s in self.slots]

    def launch(self, stream: Optional[torch.cuda.Stream] = None):
        self.big.launch(stream)
        return self.out

    def check(self) -> None:
        self.big.check()

    def update_member(self, m: int, inp, check_lengths: bool = True):
        """New inputs for request m (same shapes; sentence lengths <= the T the slot was captured with - the longest sentence of
        the requests it was created from; capture with padded lengths when they vary)."""
        lens = inp.get("srl_arg_word_mask_len") if check_lengths else None
        if lens is not None:
            mx = int((lens if isinstance(lens, torch.Tensor) else torch.as_tensor(lens)).max())
            if mx > self.big.T:
                raise ValueError(f"request has a sentence of {mx} tokens but this slot was captured with T = {self.big.T}")
        view = self.slots[m].inp
        for k, v in inp.items():
            if k in view:
                view[k].copy_(v if isinstance(v, torch.Tensor) else torch.from_numpy(v), non_blocking=True)


class Group:
    """G batch slots + one shared language-encoder workspace + one graph for all."""

    def __init__(self, eng: VogEngine, inps, with_pred=True, graph=True, pred_rec=None):
        # pred_rec: optional [len(inps) * B, record_words] buffer; member m writes rows [m*B, (m+1)*B)
        assert len(inps) >= 1
        self.eng = eng
        dev = eng.device
        d = eng.desc

        def dev_t(v):
            return (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).to(dev)

        # the members' word-level arrays live back to back: the group encoder reads them as ONE batch
        self.lang_in = {k: torch.cat([dev_t(i[k]) for i in inps], dim=0).contiguous() for k in LANG_KEYS}
        sizes = [int(dev_t(i["srl_arg_words_ind"]).shape[0]) for i in inps]
        assert len(set(sizes)) == 1, "group members must have the same batch size"
        B = sizes[0]
        self.slots = []
        with torch.cuda.device(dev):
            T_max = int(self.lang_in["srl_arg_word_mask_len"].max().item())
            for m, inp in enumerate(inps):
                mi = {k: dev_t(v) for k, v in inp.items() if k not in LANG_KEYS}
                for k in LANG_KEYS:
                    mi[k] = self.lang_in[k][m * B:(m + 1) * B]
                self.slots.append(Slot(eng, mi, T_max, with_pred, False,
                                       None if pred_rec is None else pred_rec[m * B:(m + 1) * B]))
            s0 = self.slots[0]
            self.B_total, self.ncmp, self.T = B * len(inps), s0.ncmp, T_max
            n = eng.lib.vog_lang_workspace_bytes(eng.ctx, self.B_total, self.ncmp, self.T)
            self.lang_ws = torch.empty(int(n), dtype=torch.uint8, device=dev)
            L.check(eng.lib.vog_lang_workspace_init(eng.ctx, self.B_total, self.ncmp, self.T,
                                                    self.lang_ws.data_ptr(), self.lang_ws.numel(), L.stream_ptr()),
                    "vog_lang_workspace_init")
            self.lb = L.Batch()
            self._lb_fault = torch.zeros(16, dtype=torch.int32).pin_memory()
            self._lb_fault_seen = 0
            self.lb.fault = self._lb_fault.data_ptr()
            self.lb.B, self.lb.ncmp, self.lb.T = self.B_total, self.ncmp, self.T
            for k in LANG_KEYS:
                setattr(self.lb, k, L.ptr(self.lang_in[k]))
            lang, fh = C.c_void_p(), C.c_void_p()
            L.check(eng.lib.vog_lang_outputs(eng.ctx, self.B_total, self.ncmp, self.T, self.lang_ws.data_ptr(),
                                             C.byref(lang), C.byref(fh)), "vog_lang_outputs")
            nvl = self.ncmp if eng.sep else 1
            for m, sl in enumerate(self.slots):
                sl.batch.shared_lang = lang.value + m * B * nvl * d.nsrl * d.lang_enc * 4
                sl.batch.shared_final_hidden = fh.value + m * B * nvl * d.lang_enc * 4
            G = len(self.slots)
            self._members = (C.POINTER(L.Batch) * G)(*[C.pointer(sl.batch) for sl in self.slots])
            self._wss = (C.c_void_p * G)(*[sl.ws.data_ptr() for sl in self.slots])
            self._wsb = (C.c_size_t * G)(*[sl.ws.numel() for sl in self.slots])
            self.graph = None
            torch.cuda.synchronize()
            if graph:
                cap = torch.cuda.Stream(device=dev)
                g = C.c_void_p()
                L.check(eng.lib.vog_group_graph_capture(eng.ctx, C.byref(self.lb), self.lang_ws.data_ptr(),
                                                        self.lang_ws.numel(), self._members, self._wss, self._wsb,
                                                        G, cap.cuda_stream, C.byref(g)), "vog_group_graph_capture")
                self.graph = g
                torch.cuda.synchronize()

    @property
    def out(self):
        return [s.out for s in self.slots]

    def launch(self, stream: Optional[torch.cuda.Stream] = None):
        for sl in self.slots:
            sl._check_epoch()
            sl.check()
        if self._lb_fault_seen != int(self._lb_fault[0]):
            k, self._lb_fault_seen = int(self._lb_fault[0]) - self._lb_fault_seen, int(self._lb_fault[0])
            self.eng._stalled(k, "a group's shared language encoder")
        _lane_enter(self.eng.device, stream)
        sp = L.stream_ptr(stream)
        if self.eng.precise is not None:
            # fp32 path: nothing of the 16-bit group forward is needed - its persistent BiLSTM layers could only
            # stall, its outputs would be overwritten
            for sl in self.slots:
                sl._precise(stream)
            return self.out
        if self.graph is not None:
            L.check(self.eng.lib.vog_graph_launch(self.graph, sp), "vog_graph_launch")
        else:
            L.check(self.eng.lib.vog_group_forward(self.eng.ctx, C.byref(self.lb), self.lang_ws.data_ptr(),
                                                   self.lang_ws.numel(), self._members, self._wss, self._wsb,
                                                   len(self.slots), sp), "vog_group_forward")
        return self.out

    def __del__(self):
        try:
            if self.graph is not None:
                self.eng.lib.vog_graph_destroy(self.graph)
        except Exception:
            pass


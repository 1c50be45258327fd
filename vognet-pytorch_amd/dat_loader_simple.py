"""Device-side SPAT / TEMP batch assembly — the step in front of the forward path.

The reference builds a SPAT / TEMP sample on the CPU inside the dataset
(`AV_CS.verb_item_getter_SPAT / _TEMP`, code/dat_loader_simple.py:1046-1338): four per-video items are
shifted (x += 720 * video / frame += 10 * video), re-ordered and concatenated, one query at a time, and
the collated batch is copied to the GPU. Here the per-video items of a whole batch (what
`AV_CS.itemcollector` stacks, [B, ncmp, ...]) are handed over as they are and `vog_assemble_batch`
(csrc/assemble.hip) writes the forward's / loss's tensors straight into their device buffers - e.g. a
`Slot`'s persistent inputs. Dataset reading itself (h5 / csv files, the 530 GB dataset) stays out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L

FWD_KEYS = ("pad_proposals", "pad_region_feature", "seg_feature_for_frms")


class PackedStaging:
    """Host -> device staging of a batch as ONE copy. The tensors of a batch (per-video items for the assembler, word-level
    language arrays, ...) live back to back, 256-byte aligned, in ONE pinned host buffer and in one device buffer of the same
    layout; `host[k]` / `dev[k]` are views. `upload()` is a single asynchronous H2D copy of the used bytes on the current
    stream (13 per-key `copy_` calls measured 8.9 GB/s on the driver's box in round 3: every call pays its own launch and
    its own sub-MB transfer; the link wants one large one). The loader side fills `host[k]` in place (`fill`, or writes
    straight into the views), so nothing is concatenated on the host."""

    def __init__(self, spec: Dict[str, torch.Tensor], device: Optional[torch.device] = None, n_dev: int = 1):
        """spec: name -> example tensor / array (shape and dtype are taken from it; its values are copied in).
        n_dev = 2: two device buffers used alternately (`upload` flips), so that the copy of batch i + 1 - issued on a
        copy stream - runs while the forward of batch i still reads its own (`upload_on`)."""
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        ex = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in spec.items()}
        self.layout, off = {}, 0
        for k, v in ex.items():
            nb = v.numel() * v.element_size()
            self.layout[k] = (off, nb, tuple(v.shape), v.dtype)
            off += (nb + 255) // 256 * 256
        self.nbytes = off
        self.hbuf = torch.empty(off, dtype=torch.uint8).pin_memory()
        self.dbufs = [torch.empty(off, dtype=torch.uint8, device=self.device) for _ in range(max(1, int(n_dev)))]
        self.host = {k: self.hbuf[o:o + nb].view(dt).view(*shp) for k, (o, nb, shp, dt) in self.layout.items()}
        self.devs = [{k: b[o:o + nb].view(dt).view(*shp) for k, (o, nb, shp, dt) in self.layout.items()} for b in self.dbufs]
        self.dbuf, self.dev = self.dbufs[0], self.devs[0]
        self._cur = 0
        self._ready = [None] * len(self.dbufs)       # event: the copy into buffer b has landed
        self._free = [None] * len(self.dbufs)        # event: the consumer of buffer b is done reading it
        self.fill(ex)

    def fill(self, items: Dict[str, torch.Tensor]) -> "PackedStaging":
        for k, v in items.items():
            if k in self.host:
                self.host[k].copy_(v if isinstance(v, torch.Tensor) else torch.from_numpy(v))
        return self

    def upload(self) -> Dict[str, torch.Tensor]:
        """ONE async copy on the current stream; returns the device views (valid once the stream reaches this point)."""
        b = self._cur
        self._cur = (b + 1) % len(self.dbufs)
        self.dbufs[b].copy_(self.hbuf, non_blocking=True)
        return self.devs[b]

    def upload_on(self, copy_stream: "torch.cuda.Stream", consumer: Optional["torch.cuda.Stream"] = None):
        """The same copy on `copy_stream` (so it overlaps whatever the consumer stream is still running - a copy issued on
        the forward's own stream waits for the previous forward there: 11.5 k instead of 26 k queries/s at cfg 2), into the
        next device buffer. The consumer stream (default: the current one) waits for the copy; call `release()` on it once
        the kernels that read the views are enqueued, so the buffer's next copy waits for them. Returns the device views."""
        b = self._cur
        self._cur = (b + 1) % len(self.dbufs)
        cons = consumer if consumer is not None else torch.cuda.current_stream(self.device)
        if self._free[b] is not None:
            copy_stream.wait_event(self._free[b])
        with torch.cuda.stream(copy_stream):
            self.dbufs[b].copy_(self.hbuf, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        self._ready[b] = ev
        cons.wait_event(ev)
        self._last = b
        return self.devs[b]

    def release(self, consumer: Optional["torch.cuda.Stream"] = None) -> None:
        cons = consumer if consumer is not None else torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(cons)
        self._free[self._last] = ev


class DeviceBatchAssembler:
    def __init__(self, cfg, comm):
        self.conc_type = cfg.ds.conc_type
        assert self.conc_type in ("spat", "temp"), "sep / svsq batches need no assembly (verb_item_getter_SEP)"
        self.nfrm0 = int(cfg.ds.num_sampled_frm)
        self.nppf0 = int(comm["num_prop_per_frm"])
        self.vid_w = float(cfg.ds.resized_width)
        self.lib = L.load()

    def __call__(self, items: Dict[str, torch.Tensor], out: Optional[Dict[str, torch.Tensor]] = None,
                 with_loss_keys: bool = True) -> Dict[str, torch.Tensor]:
        """items: device tensors with leading axes [B, ncmp]. `out`: optional existing destination tensors
        (e.g. `slot.inp`) for any of the produced keys; missing ones are allocated."""
        P = items["pad_proposals"]
        assert P.is_cuda and P.dtype == torch.float32 and P.dim() == 4 and P.shape[-1] == 7
        B, ncmp, NPv, _ = P.shape
        assert NPv == self.nfrm0 * self.nppf0
        dev = P.device
        R, S = items["pad_region_feature"], items["seg_feature_for_frms"]
        out = dict(out or {})

        def dst(k, shape, dtype):
            t = out.get(k)
            if t is None:
                t = torch.empty(shape, dtype=dtype, device=dev)
                out[k] = t
            assert tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_cuda and t.is_contiguous(), k
            return t

        a = L.AssembleArgs()
        keep = [P.contiguous(), R.contiguous(), S.contiguous()]
        a.props_in, a.region_in, a.seg_in = (L.ptr(t) for t in keep)
        a.props_out = L.ptr(dst("pad_proposals", (B, ncmp * NPv, 7), torch.float32))
        a.region_out = L.ptr(dst("pad_region_feature", (B, ncmp * NPv, R.shape[-1]), torch.float32))
        a.seg_out = L.ptr(dst("seg_feature_for_frms", (B, ncmp * self.nfrm0, S.shape[-1]), torch.float32))
        if "pad_pnt_mask" in items:
            pm = items["pad_pnt_mask"].to(torch.uint8).contiguous()
            keep.append(pm)
            a.pnt_in, a.pnt_out = L.ptr(pm), L.ptr(dst("pad_pnt_mask", (B, ncmp * NPv), torch.uint8))
        if with_loss_keys and "pad_gt_bboxs" in items:
            G = items["pad_gt_bboxs"].shape[2]
            sb = items["srl_boxes"]
            for k in ("pad_gt_bboxs", "num_box", "target_cmp", "srl_boxes", "srl_boxes_lens"):
                keep.append(items[k].contiguous())
            a.gt_in, a.num_box, a.target_cmp, a.srl_boxes_in, a.srl_boxes_lens = (L.ptr(t) for t in keep[-5:])
            a.gt_out = L.ptr(dst("pad_gt_bboxs", (B, G, 5), torch.float32))
            a.num_box_out = L.ptr(dst("num_box", (B,), torch.int64))
            a.srl_boxes_out = L.ptr(dst("srl_boxes", tuple(sb.shape), torch.int64))
            a.frm_out = L.ptr(dst("pad_frm_mask", (B, ncmp * NPv, G), torch.uint8))
            a.G, a.nv, a.nsrl, a.nbox = G, sb.shape[1], sb.shape[2], sb.shape[3]
        a.B, a.ncmp, a.nfrm0, a.nppf0 = B, ncmp, self.nfrm0, self.nppf0
        a.prop_dim, a.seg_dim = R.shape[-1], S.shape[-1]
        a.conc_type, a.vid_w = L.CONC_TYPE[self.conc_type], self.vid_w
        L.check(self.lib.vog_assemble_batch(C.byref(a), L.stream_ptr()), "vog_assemble_batch")
        out["_keepalive"] = keep
        return out

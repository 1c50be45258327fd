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


class DevicePrefetcher:
    """`for dev_batch, host_batch in DevicePrefetcher(loader, device, depth, hold)`: the loader's batches (dicts of CPU tensors)
    as device batches whose host -> device copies were issued `depth` batches ahead on a copy stream, so that they overlap the
    forwards of the batches in front of them and the loop never waits for a copy it has just issued (the reference's
    `batch[k].to(device)` per key and batch, code/utils/trn_utils.py:478 / :562, is a blocking copy per key: 1.2 ms per
    cfg-2 batch, 6 forwards' worth).

    The copies land in persistent device buffers (a ring of depth + hold + 1 sets per batch shape; a batch of another shape -
    the short tail batch - gets a set of its own); a set is rewritten only after the consumer's stream has passed the point
    where it gave the batch back: the consumer may keep using the last `hold` batches it was handed (dynamic batching
    concatenates `hold` of them), nothing older. Batches that already live on the device pass through untouched.
    Pinned host tensors make the copies asynchronous; pageable ones still work (staged by the runtime)."""

    def __init__(self, loader, device, depth: int = 2, hold: int = 1):
        self.loader, self.device = loader, torch.device(device)
        self.depth, self.hold = max(1, int(depth)), max(1, int(hold))

    def __iter__(self):
        dev = self.device
        if dev.type != "cuda":
            for bt in self.loader:
                yield bt, bt
            return
        from collections import deque
        cs = torch.cuda.Stream(device=dev)
        nring = self.depth + self.hold + 1
        ring: list = [None] * nring
        given = {}                                   # batch number -> event on the consumer's stream behind its use
        pending = deque()
        n = 0

        SMALL = 64 << 10        # tensors below this travel together: one pinned buffer, ONE transfer (a transfer costs ~8 us of
                                # copy-engine time whatever its size: 20 KB-sized arrays per batch were 40 % of a cfg-2 batch's copies)

        def make_set(bt, sig):
            small = [k for k, v in bt.items() if v.numel() * v.element_size() < SMALL]
            e = {"sig": sig, "bufs": {k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in bt.items() if k not in small},
                 "small": small}
            if small:
                off, lay = 0, {}
                for k in small:
                    v = bt[k]
                    nb = v.numel() * v.element_size()
                    lay[k] = (off, nb)
                    off += (nb + 255) // 256 * 256
                e["hpack"] = [torch.empty(off, dtype=torch.uint8).pin_memory() for _ in range(2)]     # filled alternately:
                e["hfree"] = [None, None]                                                             # a buffer is rewritten once its transfer is done
                e["dpack"] = torch.empty(off, dtype=torch.uint8, device=dev)
                e["hviews"] = [{k: h[o:o + nb].view(bt[k].dtype).view(bt[k].shape) for k, (o, nb) in lay.items()} for h in e["hpack"]]
                for k, (o, nb) in lay.items():
                    e["bufs"][k] = e["dpack"][o:o + nb].view(bt[k].dtype).view(bt[k].shape)
                e["turn"] = 0
            e["bufs"] = {k: e["bufs"][k] for k in bt}           # the loader's key order
            return e

        def issue(bt):
            nonlocal n
            if any(v.is_cuda for v in bt.values()):
                pending.append((bt, bt, None))
                n += 1
                return
            sig = tuple((k, tuple(v.shape), v.dtype) for k, v in bt.items())
            e = ring[n % nring]
            cur = torch.cuda.current_stream(dev)
            if e is None or e["sig"] != sig:
                # new buffers come from the consumer stream's pool: whatever used that memory before is ordered on that stream
                e = make_set(bt, sig)
                ring[n % nring] = e
                ev = torch.cuda.Event()
                ev.record(cur)
                cs.wait_event(ev)
            else:
                m = n - nring + self.hold - 1          # the batch whose hand-back frees this set
                if m in given:
                    cs.wait_event(given[m])
            small = e["small"]
            if small:
                t = e["turn"]
                e["turn"] = 1 - t
                if e["hfree"][t] is not None:
                    e["hfree"][t].synchronize()        # (two uses of this set ago: long done)
                hv = e["hviews"][t]
                for k in small:
                    hv[k].copy_(bt[k])
            with torch.cuda.stream(cs):
                for k, v in bt.items():
                    if not small or k not in hv:
                        e["bufs"][k].copy_(v, non_blocking=True)
                if small:
                    e["dpack"].copy_(e["hpack"][t], non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(cs)
                if small:
                    e["hfree"][t] = ready
            pending.append((e["bufs"], bt, ready))
            n += 1

        def hand_out():
            dbt, hbt, ready = pending.popleft()
            if ready is not None:
                torch.cuda.current_stream(dev).wait_event(ready)
            return dbt, hbt

        j = 0
        for bt in self.loader:
            issue(bt)
            if len(pending) > self.depth:
                yield hand_out()
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                given[j] = ev
                given.pop(j - 2 * nring, None)
                j += 1
        while pending:
            yield hand_out()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            given[j] = ev
            j += 1


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
        a, out = self.args(items, out, with_loss_keys)
        L.check(self.lib.vog_assemble_batch(C.byref(a), L.stream_ptr()), "vog_assemble_batch")
        return out

    def args(self, items: Dict[str, torch.Tensor], out: Optional[Dict[str, torch.Tensor]] = None,
             with_loss_keys: bool = True):
        """The vog_assemble_args of this call and the destination dict, without launching (a fed slot captures the launch
        into its graph: `engine.Slot.feed_from`).
        items: device tensors with leading axes [B, ncmp] - or PINNED host tensors (zero copy: pinned memory is mapped
        into the device's address space, the kernels read it over the host link, so the batch needs no DMA of its own and
        none of a copy's fixed latency; `out` must then name the destination tensors, which also fixes the device).
        `out`: optional existing destination tensors (e.g. `slot.inp`) for any of the produced keys; missing ones are
        allocated."""
        P = items["pad_proposals"]
        assert (P.is_cuda or P.is_pinned()) and P.dtype == torch.float32 and P.dim() == 4 and P.shape[-1] == 7
        B, ncmp, NPv, _ = P.shape
        assert NPv == self.nfrm0 * self.nppf0
        if P.is_cuda:
            dev = P.device
        else:
            assert out and "pad_region_feature" in out, "pinned host items: pass the destination tensors in `out`"
            dev = out["pad_region_feature"].device
            assert all(items[k].is_pinned() for k in ("pad_region_feature", "seg_feature_for_frms")), "host items must be pinned"
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
        out["_keepalive"] = keep
        return a, out

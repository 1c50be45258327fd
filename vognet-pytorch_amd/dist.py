"""Multi-GPU glue: one process per GPU, queries sharded like the reference's
`NewDistributedSampler` (utils/trn_utils.py:127-156: contiguous
`num_samples * rank` slices), and ONE collective on the data path — an
all-gather of the packed per-query prediction records over RCCL/xGMI (backend
"nccl" is RCCL on ROCm) — replacing the reference's pickle-file gather
(code/eval_vsrl_corr.py:125-140). Result order is rank-major, i.e. the order
rank 0 gets by concatenating ranks 0..W-1 in the reference.

Records are 6.8 KB/query: the exchange is latency-bound, so it is issued as a
single `all_gather_into_tensor` of the whole per-rank buffer (one launch, all
seven xGMI links used by RCCL's direct algorithm) rather than per-tensor calls.
Works on CPU tensors with the gloo backend (tests, world_size 2).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def synchronize():
    if is_dist() and get_world_size() > 1:
        dist.barrier()


def init_from_env(backend: str = "nccl"):
    """env:// rendezvous as reference main_dist.py:107-109 (RANK/WORLD_SIZE/
    MASTER_ADDR/MASTER_PORT from torchrun)."""
    if is_dist() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")


def _parse_cpulist(txt: str):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_host_to_device_node(dev_index: int = 0):
    """Run this process on the CPUs of the NUMA node its GPU hangs off (what `numactl --cpunodebind` per rank does in a
    launcher script; torchrun does not). Pinned staging buffers are then allocated on that node (first touch) and host -> device
    transfers do not cross the socket interconnect: on the 2-socket MI355X boxes the host-fed rate of one and the same binary
    varied 17.8 - 21.7 k queries/s from run to run without it. Returns the node, or None when the topology cannot be read
    (the affinity is then left alone). Call it before anything allocates pinned memory."""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bus = "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of this rank's contiguous slice; every rank gets
    ceil(n/world) items, the tail wraps around to the first items
    (DistributedSampler padding, trn_utils.py:147-153)."""
    per = (n + world - 1) // world
    return per * rank, per * (rank + 1)


def shard_indices(n: int, rank: int, world: int):
    a, b = shard_range(n, rank, world)
    return [i % n for i in range(a, b)]


def all_gather_records(rec: torch.Tensor) -> torch.Tensor:
    """[b_local, rec_words] -> [world*b_local, rec_words], rank-major."""
    w = get_world_size()
    if w == 1:
        return rec
    rec = rec.contiguous()
    out = torch.empty((w * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    if dist.get_backend() == "gloo" and not hasattr(dist, "all_gather_into_tensor"):
        parts = [torch.empty_like(rec) for _ in range(w)]
        dist.all_gather(parts, rec)
        return torch.cat(parts, 0)
    try:
        dist.all_gather_into_tensor(out, rec)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(rec) for _ in range(w)]
        dist.all_gather(parts, rec)
        out = torch.cat(parts, 0)
    return out


class RecordRing:
    """The exchange step of the N > 1 path: per-batch prediction records are staged in a ring of
    two halves and ONE all-gather moves a half (`per_half` batches of `rows` records) at a time.

    Why not a collective per batch: on one MI355X a per-batch all-gather costs 12-20 us per step (the
    process group's stream is a 5th stream on 4 hardware queues and drags a forward stream behind the
    collective's dependencies); one gather per 32 batches costs 4.6 us per step. Rows inside a half
    keep the order they were pushed in; the gathered tensor is rank-major per half, i.e. rank 0's
    concatenation order in the reference (code/eval_vsrl_corr.py:125-140) within every half.

    Works for CUDA tensors (copies on the batch's own stream, collective on `gather_stream` behind
    events, settled a whole ring later) and for CPU tensors with gloo (synchronous copies): the same
    code path is exercised by tests/test_dist_gloo.py at world_size 2 and used by bench.py and
    Evaluator.forward. `on_half(gathered [world*per_half*rows, width], n_valid_batches)` is called on
    every rank when a half has been gathered (after it settled), in push order."""

    def __init__(self, rows: int, width: int, per_half: int, device, dtype=torch.float32, on_half=None):
        self.rows, self.width, self.per_half = int(rows), int(width), int(per_half)
        self.world = get_world_size()
        self.cuda = torch.device(device).type == "cuda"
        n = self.per_half * self.rows
        self.ring = [torch.zeros(n, width, dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [torch.empty(self.world * n, width, dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]                  # pending gather of a half: True (CPU, done) or a CUDA event
        self.read_done = [None, None]             # CUDA event: the last gather that READ ring[h] has finished
        self.count = [0, 0]                       # batches staged in the half when it was sent
        self.half, self.n = 0, 0
        self.on_half = on_half
        self.gather_stream = torch.cuda.Stream(device=device) if self.cuda else None
        self._streams = set()
        self.gathers = 0

    def _settle(self, h: int):
        """The gather of half h has to be complete before `on_half` reads `gathered[h]`: on CUDA the
        CURRENT stream waits for the event recorded behind the gather on `gather_stream` (the
        single-rank copy and the RCCL collective alike), so the reads `on_half` enqueues are ordered."""
        if self.work[h] is not None:
            if self.cuda:
                torch.cuda.current_stream().wait_event(self.work[h])
            self.work[h] = None
            if self.on_half is not None:
                self.on_half(self.gathered[h], self.count[h])

    def push(self, rec: torch.Tensor, stream=None):
        """Stage one batch's records ([rows, width]); sends the half when it is full."""
        h, n = self.half, self.n
        if n == 0:
            self._settle(h)                        # this half was sent a whole ring ago: never stalls in steady state
        dst = self.ring[h][n * self.rows:(n + 1) * self.rows]
        if self.cuda:
            st = stream if stream is not None else torch.cuda.current_stream()
            if self.read_done[h] is not None:      # the previous gather of this half may still be reading ring[h]
                st.wait_event(self.read_done[h])
            with torch.cuda.stream(st):
                dst.copy_(rec, non_blocking=True)
            self._streams.add(st)
        else:
            dst.copy_(rec)
        self.n = n + 1
        if self.n == self.per_half:
            self._send(st if self.cuda else None)

    def _send(self, pushing_stream=None):
        h = self.half
        self.count[h] = self.n
        if self.world > 1 or self.on_half is not None:
            if self.cuda:
                # Multi-rank: the collective is issued IN the stream that staged the half's last rows (with 4 forward streams
                # that is always the same one: the last lane). A stream of its own is a 5th stream on 4 hardware queues: while
                # the gather waits for the other lanes' in-flight forwards it blocks the queue it shares with one of them
                # (measured with the policy forced on one GPU: 0.92 of the no-exchange rate; in the lane's own stream only
                # that lane waits, once per half).
                multi = self.world > 1 or os.environ.get("VOG_FORCE_MULTI_RANK_LANES") == "1"
                gs = pushing_stream if (multi and pushing_stream is not None) else self.gather_stream
                for s in self._streams:            # the collective runs behind every stream that staged rows
                    if s.cuda_stream == gs.cuda_stream:
                        continue
                    ev = torch.cuda.Event()
                    ev.record(s)
                    gs.wait_event(ev)
                self._streams = set()
                with torch.cuda.stream(gs):
                    self._gather(h)
                    done = torch.cuda.Event()
                    done.record(gs)
                self.work[h] = done
                self.read_done[h] = done
                if self.world > 1 or os.environ.get("VOG_FORCE_MULTI_RANK_LANES") == "1":
                    dev = self.ring[h].device
                    _PENDING_COLLECTIVE[dev.index if dev.index is not None else torch.cuda.current_device()] = done
            else:
                self._gather(h)
                self.work[h] = True
            self.gathers += 1
        elif self.cuda:
            self._streams = set()
        self.half, self.n = 1 - h, 0

    def _gather(self, h: int):
        """Enqueued on `gather_stream` (CUDA) or run synchronously (CPU / gloo). The RCCL call is the
        stream-ordered form: it returns at once and `gather_stream` does not pass it before the collective
        has finished, so the event recorded behind it covers the collective."""
        if self.world == 1:
            self.gathered[h].copy_(self.ring[h])
        else:
            dist.all_gather_into_tensor(self.gathered[h], self.ring[h])

    def flush(self):
        """Send a partially filled half and settle everything (end of the loop)."""
        if self.n:
            self._send(None)
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.gather_stream)
        first = self.half                          # older half first: push order
        for h in (first, 1 - first):
            self._settle(h)


# ---- collectives and the persistent BiLSTM share the CUs ---------------------------------------------------------------
# The event behind the most recent cross-rank gather of a device (RecordRing._send). engine._lane_enter makes the LAST forward
# lane wait for it before its next forward, so a collective never shares the chip with more than 3 forwards (4 persistent
# BiLSTM layer kernels fill the 256 CUs; a resident RCCL kernel waiting for its peers would leave one of them short).
_PENDING_COLLECTIVE = {}


def pending_collective(dev_index: int):
    return _PENDING_COLLECTIVE.get(int(dev_index))


def unpack_gathered(gathered: torch.Tensor, world: int, per_half: int, rows: int, n_valid: int) -> torch.Tensor:
    """[world*per_half*rows, W] (rank-major) -> [n_valid*world*rows, W] in (batch, rank, row) order:
    batch k of every rank back to back = the order a per-batch all-gather would have produced."""
    w = gathered.view(world, per_half, rows, -1)[:, :n_valid]
    return w.permute(1, 0, 2, 3).reshape(n_valid * world * rows, -1)


def all_reduce_grads_begin(grads, bucket_bytes: int = 32 << 20, average: bool = True):
    """Start the data-parallel gradient exchange of the training path (the reference wraps the model in
    DistributedDataParallel, code/main_dist.py:72-85) and return `finish()`: the gradient tensors of `grads` (name ->
    tensor; keys starting with '_' are skipped) are flattened IN NAME ORDER into buckets of at most `bucket_bytes` and
    every bucket is ONE asynchronous in-place all-reduce (RCCL over xGMI with backend "nccl", gloo on CPU). xGMI is
    point to point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound: few large buckets (default 32
    MiB: 177 MB of fp32 parameters = 6 collectives) instead of one per parameter. Nothing waits here: the caller goes on
    computing (the trainer starts the exchange of the visual side's gradients and runs the language side's backward
    meanwhile - the overlap DDP gets from its autograd hooks). `finish()` waits, divides by the world size, copies the
    buckets back into the tensors of `grads` and returns the number of collectives."""
    names = sorted(k for k in grads if not k.startswith("_") and isinstance(grads[k], torch.Tensor))
    w = get_world_size()
    if w == 1 or not names:
        return lambda: 0
    works, buckets, cur, cur_bytes = [], [], [], 0
    for n in names:
        t = grads[n]
        nb = t.numel() * t.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or t.dtype != grads[cur[0]].dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(n)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    flats = []
    for b in buckets:
        flat = torch.cat([grads[n].reshape(-1) for n in b])
        flats.append(flat)
        works.append(dist.all_reduce(flat, async_op=True))

    def finish():
        for b, flat, wk in zip(buckets, flats, works):
            wk.wait()
            if average:
                flat.div_(w)
            off = 0
            for n in b:
                k = grads[n].numel()
                grads[n].copy_(flat[off:off + k].view_as(grads[n]))
                off += k
        return len(buckets)
    return finish


def all_reduce_grads(grads, bucket_bytes: int = 32 << 20, average: bool = True):
    """`all_reduce_grads_begin(...)()`: the whole exchange, blocking. Returns the number of collectives issued."""
    return all_reduce_grads_begin(grads, bucket_bytes, average)()

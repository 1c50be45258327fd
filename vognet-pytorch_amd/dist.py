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

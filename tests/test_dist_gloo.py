"""world_size-2 gloo run of the N>1 path on CPU: rank-contiguous sharding and
the single all-gather of packed prediction records (rank-major order, i.e. what
rank 0 gets by concatenating per-rank results in reference eval_vsrl_corr.py:130-137)."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

D = importlib.import_module("vognet-pytorch_amd.dist")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_queries, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = D.shard_indices(n_queries, rank, world)
    # record of query i = [i, i+0.5, ...] (1700 fp32 words = 6800 B, the gt5 record)
    rec = torch.stack([torch.full((1700,), float(i)) + torch.arange(1700) * 1e-3 for i in idx])
    allr = D.all_gather_records(rec)
    meta = D.all_gather_records(torch.tensor(idx, dtype=torch.int64).view(-1, 1))
    if rank == 0:
        q.put((allr[:, 0].tolist(), meta.view(-1).tolist()))
    D.synchronize()
    dist.destroy_process_group()


def test_allgather_records_world2():
    world, n = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    first, meta = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = D.shard_indices(n, 0, world) + D.shard_indices(n, 1, world)
    assert meta == expect == [0, 1, 2, 3, 4, 5, 6, 0]
    assert [int(x) for x in first] == expect


def test_single_process_passthrough():
    x = torch.randn(4, 1700)
    assert D.all_gather_records(x) is x
    assert D.shard_range(8, 1, 2) == (4, 8)


def _worker_async(rank, world, port, q):
    """The exchange pattern of bench.py's N > 1 loop: several slots in flight, one asynchronous
    all-gather of a slot's records per step, settled (work.wait()) only before the slot is reused."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nslots, B, W = 4, 4, 1700
    recs = [torch.zeros(B, W) for _ in range(nslots)]
    gathered = [torch.zeros(world * B, W) for _ in range(nslots)]
    pending = [None] * nslots
    seen = []
    for step in range(10):
        u = step % nslots
        if pending[u] is not None:
            pending[u].wait()
            seen.append(gathered[u][:, 0].clone())
            pending[u] = None
        recs[u].fill_(float(100 * step + rank))            # "forward" of this step writes the records
        pending[u] = dist.all_gather_into_tensor(gathered[u], recs[u], async_op=True)
    for u in range(nslots):
        if pending[u] is not None:
            pending[u].wait()
            seen.append(gathered[u][:, 0].clone())
    if rank == 0:
        q.put([s.tolist() for s in seen])
    dist.barrier()
    dist.destroy_process_group()


def test_async_gather_settled_before_slot_reuse_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_async, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    seen = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(seen) == 10
    steps = sorted(int(s[0]) // 100 for s in seen)
    assert steps == list(range(10))
    for s in seen:                       # rank-major: rank 0's rows then rank 1's, same step
        st = int(s[0]) // 100
        assert s == [float(100 * st)] * 4 + [float(100 * st + 1)] * 4


def _worker_ring(rank, world, port, q):
    """bench.py's / Evaluator.forward's exchange code itself (dist.RecordRing), gloo + CPU tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, width, per_half = 4, 1700, 3
    got = []

    def on_half(g, n_valid):
        got.append(D.unpack_gathered(g, world, per_half, rows, n_valid)[:, 0].clone())

    ring = D.RecordRing(rows, width, per_half, "cpu", on_half=on_half)
    for step in range(8):                              # 8 batches: two full halves + a partial one
        ring.push(torch.full((rows, width), float(100 * step + rank)))
    ring.flush()
    assert ring.gathers == 3
    if rank == 0:
        q.put(torch.cat(got).tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_record_ring_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_ring, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    seen = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    # (batch, rank, row) order over all 8 batches: what per-batch all-gathers would have produced
    expect = [float(100 * st + r) for st in range(8) for r in range(2) for _ in range(4)]
    assert seen == expect


def test_record_ring_single_process():
    got = []
    ring = D.RecordRing(2, 5, 2, "cpu", on_half=lambda g, n: got.append(D.unpack_gathered(g, 1, 2, 2, n).clone()))
    for k in range(3):
        ring.push(torch.full((2, 5), float(k)))
    ring.flush()
    assert torch.cat(got)[:, 0].tolist() == [0.0, 0.0, 1.0, 1.0, 2.0, 2.0]


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = {"lin2.0.weight": torch.randn(8, 12, generator=g), "lin2.0.bias": torch.randn(8, generator=g),
             "mult_txf.encoder.layers.0.selfattn.layer.wo.weight": torch.randn(12, 12, generator=g),
             "_d_x": torch.full((3,), float(rank))}
    # two groups in flight at once, as the trainer issues them (visual side first, language side behind it)
    ga = {k: v for k, v in grads.items() if k.startswith("lin2") or k.startswith("_")}
    gb = {k: v for k, v in grads.items() if k.startswith("mult")}
    fa = D.all_reduce_grads_begin(ga, bucket_bytes=400)      # small buckets: more than one collective
    fb = D.all_reduce_grads_begin(gb, bucket_bytes=400)
    n = fa() + fb()
    if rank == 0:
        q.put((n, {k: v.numpy().copy() for k, v in grads.items()}))      # numpy: a tensor would travel as an fd the exiting worker may close first
    D.synchronize()
    dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """dist.all_reduce_grads (the DDP exchange of the training path): bucketed, averaged, '_' keys untouched."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    n, got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n >= 2
    ref = {}
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        for k, shape in (("lin2.0.weight", (8, 12)), ("lin2.0.bias", (8,)),
                         ("mult_txf.encoder.layers.0.selfattn.layer.wo.weight", (12, 12))):
            ref[k] = ref.get(k, 0) + torch.randn(*shape, generator=g) / world
    for k in ref:
        assert torch.allclose(torch.from_numpy(got[k]), ref[k], atol=1e-6), k
    assert torch.equal(torch.from_numpy(got["_d_x"]), torch.zeros(3))                      # rank 0's own, not reduced


# ---- world = 8 (round 5): exactly the exchange code `bench.py --gpus 8` and `Evaluator.forward` run, sized as they size it ----
def _spawn(world, target, args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in ps:
        p.start()
    got = q.get(timeout=timeout)
    for p in ps:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return got


def _worker_ring8(rank, world, port, n_units, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py::measure with 4 lanes: per_half = max(nunits, (64 // nunits) * nunits) = 64 unit launches per half ring,
    # unit_rows = G * B = 4 rows of 1700 words
    nunits, rows, width = 4, 4, 1700
    per_half = max(nunits, (64 // nunits) * nunits)
    got = []
    ring = D.RecordRing(rows, width, per_half, "cpu",
                        on_half=lambda g, n: got.append(D.unpack_gathered(g, world, per_half, rows, n)[:, :2].clone()))
    for step in range(n_units):
        rec = torch.zeros(rows, width)
        rec[:, 0] = float(step)
        rec[:, 1] = float(rank) * 10 + torch.arange(rows)
        ring.push(rec)
    ring.flush()
    assert ring.gathers == (n_units + per_half - 1) // per_half
    if rank == 0:
        q.put(torch.cat(got).tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_record_ring_world8_bench_geometry():
    """Two full half rings + a partial one on 8 ranks, `per_half` as bench.py computes it for its 4 lanes: rank 0 sees every
    unit's rows in (unit, rank, row) order - what a gather per unit would have produced."""
    world, n_units = 8, 2 * 64 + 22
    seen = _spawn(world, _worker_ring8, (n_units,))
    expect = [[float(st), float(r * 10 + i)] for st in range(n_units) for r in range(world) for i in range(4)]
    assert seen == expect


class _FakeModel(torch.nn.Module):
    """Stands in for the HIP model on CPU: records that carry the query's ann_idx, so that the merge order is visible."""
    supports_T_hint = False

    def __init__(self, rw, nsrl, NP):
        super().__init__()
        self.rw, self.nsrl, self.NP = rw, nsrl, NP

    def forward(self, batch):
        B = batch["ann_idx"].shape[0]
        rec = torch.zeros(B, self.rw)
        rec[:, 0] = batch["ann_idx"].float()
        return {"mdl_outs_eval": torch.zeros(B, 1, self.nsrl, self.NP), "_pred_rec": rec}


def _worker_eval8(rank, world, port, n_batches, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from oracle import cases
    ev_mod = importlib.import_module("vognet-pytorch_amd.eval_vsrl_corr")
    cfg, sd, batch, c = cases.build("small/vog_spat")
    cfg.train.bsv = 2
    comm = {"vocab_size": c["vocab"], "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": c["nppf0"]}
    evl = ev_mod.EvaluatorSPAT(cfg, comm, torch.device("cpu"))
    ncmp, nsrl, nfrm0 = 4, 5, 10
    rw = nsrl * ncmp * nfrm0 * 8 + nsrl * nfrm0 * 2
    B = 2
    dl = []
    for i in D.shard_indices(n_batches, rank, world):
        b = 1 if i == n_batches - 1 else B                       # the loader's LAST batch is short (drop_last = False)
        ids = np.arange(i * B, i * B + b, dtype=np.int64)
        dl.append({"ann_idx": torch.from_numpy(ids), "sent_idx": torch.from_numpy(ids.copy()),
                   "new_srl_idxs": torch.zeros(b, ncmp, dtype=torch.int64), "num_cmp_msk": torch.ones(b, ncmp, dtype=torch.int64),
                   "target_cmp": torch.zeros(b, dtype=torch.int64), "permute": torch.arange(ncmp).repeat(b, 1),
                   "permute_inv": torch.arange(ncmp).repeat(b, 1)})
    mdl = _FakeModel(rw, nsrl, 200)
    evl(mdl, None, dl, "valid", rank=rank, pred_path=tmp)
    if rank == 0:
        q.put("done")
    dist.barrier()
    dist.destroy_process_group()


def test_evaluator_forward_world8_rank_major_merge(tmp_path):
    """`Evaluator.forward` itself on 8 gloo ranks (CPU, a stand-in model): 21 loader batches of 2 queries, the last one short;
    `shard_indices` gives every rank 3 batches (the tail wraps: rank 7 owns [21 -> 0, 1, 2]); the short batch belongs to a
    middle rank (6). Rank 0's pickle = all of rank 0's queries, then rank 1's, ... - the order in which the reference appends
    its per-rank files (code/eval_vsrl_corr.py:131-137) - with the short batch's padding rows dropped."""
    import pickle
    world, n_batches = 8, 21
    _spawn(world, _worker_eval8, (n_batches, str(tmp_path)))
    recs = pickle.load(open(tmp_path / "valid_0.pkl", "rb"))
    expect = []
    for r in range(world):
        for i in D.shard_indices(n_batches, r, world):
            expect += [2 * i] if i == n_batches - 1 else [2 * i, 2 * i + 1]
    assert D.shard_indices(n_batches, 6, world) == [18, 19, 20] and D.shard_indices(n_batches, 7, world) == [0, 1, 2]
    assert [r["idx_vid"] for r in recs] == expect
    assert [int(r["pred_boxes"][0][0][0][0]) for r in recs] == expect      # the records travelled with their metadata

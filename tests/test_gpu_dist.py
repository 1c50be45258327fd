"""Two ranks with REAL device kernels (one MI355X, both processes on cuda:0, gloo collectives on device tensors): the
closest this environment gets to the N > 1 path. RCCL needs one GPU per rank, so the collective backend here is gloo;
everything else - the engine, the record ring, the trainer, the gradient exchange code - is what a multi-GPU run executes.
Reference counterparts: the rank-0 merge of eval_vsrl_corr.py:125-140, DistributedDataParallel of main_dist.py:72-85."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(name):
    from oracle import cases
    synth = importlib.import_module("vognet-pytorch_amd.synth")
    sel_mod = importlib.import_module("vognet-pytorch_amd.mdl_selector")
    cfg, sd, batch, c = cases.build(name)
    comm = {"vocab_size": c["vocab"], "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": c["nppf0"]}
    sel = sel_mod.get_mdl_loss_eval(cfg)
    return cfg, sd, c, comm, sel, synth


def _rank_batch(synth, cfg, c, comm, rank, B):
    b = synth.make_batch(cfg.ds.conc_type, B, c["nppf0"], ncmp=c["ncmp"], vocab_size=c["vocab"], prop_dim=cfg.mdl.prop_feat_dim,
                         seg_dim=cfg.mdl.seg_feat_dim, seed=4200 + rank, ragged=True)
    b.update(synth.make_targets(b, cfg.ds.conc_type, c["nppf0"], seed=77 + rank))
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in b.items()}


def _train_worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    trn = importlib.import_module("vognet-pytorch_amd.train")
    cfg, sd, c, comm, sel, synth = _setup(name)
    tr = trn.FP32Trainer(cfg, comm, {k: torch.from_numpy(v) for k, v in sd.items()}, sel["loss"](cfg, comm), lr=1e-4)
    dev = _rank_batch(synth, cfg, c, comm, rank, 2)
    losses = [float(tr.step(dev)["loss"]) for _ in range(2)]
    torch.cuda.synchronize()
    q.put((rank, losses, {k: v.cpu().numpy() for k, v in tr.state_dict().items()}))   # (numpy: pickled by value)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_equals_training_on_the_averaged_gradient():
    """Two FP32Trainer ranks, each on its own batch, gradients averaged through `dist.all_reduce_grads_begin / finish` (the
    DDP step): after two optimisation steps both ranks hold the same parameters, and they are the parameters of ONE trainer
    stepping on the average of the two batches' gradients."""
    name, world = "small/vog_spat", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_train_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, losses, params = q.get(timeout=300)
        got[r] = (losses, params)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in got[0][1]:
        assert np.array_equal(got[0][1][k], got[1][1][k]), k           # replicas stay bit-identical
    # one process: the same two steps on the averaged gradient
    trn = importlib.import_module("vognet-pytorch_amd.train")
    L = importlib.import_module("vognet-pytorch_amd.lib")
    cfg, sd, c, comm, sel, synth = _setup(name)
    tr = trn.FP32Trainer(cfg, comm, {k: torch.from_numpy(v) for k, v in sd.items()}, sel["loss"](cfg, comm), lr=1e-4)
    batches = [_rank_batch(synth, cfg, c, comm, r, 2) for r in range(world)]
    for it in range(2):
        gs = [tr.gradients(b) for b in batches]
        for r in range(world):
            assert abs(float(gs[r][0]["loss"]) - got[r][0][it]) <= 1e-6 * abs(got[r][0][it])
        tr.num_it += 1
        tr.adam_step += 1
        for k in sorted(gs[0][1]):
            g = ((gs[0][1][k] + gs[1][1][k]) / world).contiguous()
            p = tr.params[k]
            if k not in tr.m:
                tr.m[k], tr.v[k] = torch.zeros_like(p), torch.zeros_like(p)
            L.check(tr.lib.vog_adam_f32(L.ptr(p), L.ptr(g), L.ptr(tr.m[k]), L.ptr(tr.v[k]), p.numel(), tr.lr, tr.betas[0], tr.betas[1],
                                        tr.eps, tr.adam_step, L.stream_ptr()), "vog_adam_f32")
    torch.cuda.synchronize()
    worst = 0.0
    for k, v in tr.state_dict().items():
        worst = max(worst, float(np.abs(v.cpu().numpy() - got[0][1][k]).max()))
    assert worst <= 1e-7, worst


def _eval_worker(rank, world, port, name, tmp, q, batch_requests=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module("vognet-pytorch_amd.dist")
    cfg, sd, c, comm, sel, synth = _setup(name)
    cfg.hip.batch_requests = batch_requests
    mdl = sel["mdl"](cfg=cfg, comm=comm)
    mdl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    evl = sel["eval"](cfg, comm, torch.device("cuda", 0))
    loss_fn = sel["loss"](cfg, comm)
    n_batches, B = 5, 2
    dl = []
    for i in D.shard_indices(n_batches, rank, world):
        b = synth.make_batch(cfg.ds.conc_type, B, c["nppf0"], ncmp=c["ncmp"], vocab_size=c["vocab"], prop_dim=cfg.mdl.prop_feat_dim,
                             seg_dim=cfg.mdl.seg_feat_dim, seed=900 + i)
        b.update(synth.make_targets(b, cfg.ds.conc_type, c["nppf0"], seed=i))
        ncmp = b["num_cmp_msk"].shape[1]
        b.update({"ann_idx": np.arange(i * B, (i + 1) * B, dtype=np.int64), "sent_idx": np.arange(i * B, (i + 1) * B, dtype=np.int64),
                  "permute": np.tile(np.arange(ncmp), (B, 1)).astype(np.int64), "permute_inv": np.tile(np.arange(ncmp), (B, 1)).astype(np.int64)})
        dl.append({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()})
    with torch.no_grad():
        val_loss, val_acc = evl(mdl, loss_fn, dl, "valid", rank=rank, pred_path=tmp)
    torch.cuda.synchronize()
    q.put((rank, {k: float(v) for k, v in val_loss.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch_requests", [1, 2])
def test_two_rank_evaluator_merges_records_rank_major(tmp_path, batch_requests):
    """`Evaluator.forward` on two ranks (HIP forward, device loss, record ring, cross-rank gather on device tensors): rank 0's
    pickle holds every query of both ranks' shards, rank-major = the order the reference builds from its per-rank files - with
    one forward per loader batch and with two loader batches per forward (`cfg.hip.batch_requests`, a short last group)."""
    import pickle
    name, world = "small/vog_spat", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_eval_worker, args=(r, world, port, name, tmp_path, q, batch_requests)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300)[:2] for _ in range(world))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    D = importlib.import_module("vognet-pytorch_amd.dist")
    recs = pickle.load(open(tmp_path / "valid_0.pkl", "rb"))
    expect = []
    for r in range(world):
        for i in D.shard_indices(5, r, world):
            expect += [2 * i, 2 * i + 1]
    # NewDistributedSampler pads the shorter shard by wrapping (utils/trn_utils.py:127-156): rank 1's last batch repeats batch 0
    assert [r["idx_vid"] for r in recs] == expect
    assert all(np.isfinite(v) and v > 0 for v in res[0].values())


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the environment),
    with both ranks on the one GPU of this box and gloo instead of RCCL (test hooks VOG_BENCH_DEVICE / VOG_BENCH_BACKEND): the
    N > 1 code path of the bench - record ring with a cross-rank gather, barrier + max-over-ranks timing, one JSON line from
    rank 0 only, `value` = the queries of BOTH ranks. Two streams per rank keep the persistent BiLSTM kernels of the two
    processes within the four the chip can hold."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VOG_BENCH_DEVICE="0", VOG_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "8",
                                       "--streams", "2", "--no-cpu-baseline", "--no-train-extra", "--no-cobatch-extra"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # rank 0 only
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 40 and d["scaling"] == "weak"
    assert d["parity"]["ok"] and d["parity"]["non_finite_outputs_all_slots"] == 0
    assert d["value"] == pytest.approx(2 * d["per_rank_value"]) and d["value"] > 50          # (gloo moves the records through the host: slow, functional only)
    assert d["config"]["global_batch"] == 8


def _bcast_worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    trn = importlib.import_module("vognet-pytorch_amd.train")
    cfg, sd, c, comm, sel, synth = _setup(name)
    # replicas that were NOT seeded alike: rank r starts from weights shifted by r (what torch.manual_seed(rank) does to an
    # init that draws from torch's generator)
    sd_r = {k: torch.from_numpy(v) + (0.01 * rank if np.issubdtype(v.dtype, np.floating) else 0) for k, v in sd.items()}
    tr = trn.FP32Trainer(cfg, comm, sd_r, sel["loss"](cfg, comm), lr=1e-4)
    before = float(next(iter(tr.params.values())).double().sum())
    tr.broadcast_from_rank0()
    dev = _rank_batch(synth, cfg, c, comm, rank, 2)
    tr.step(dev)
    torch.cuda.synchronize()
    q.put((rank, before, {k: v.cpu().numpy() for k, v in tr.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_initial_weights_are_broadcast_from_rank0():
    """the data-parallel path must not rely on identical seeding (DistributedDataParallel broadcasts rank 0's
    parameters at construction, code/main_dist.py:72-85). Two ranks that start from DIFFERENT weights hold rank 0's after
    `broadcast_from_rank0` and identical ones after a step on different batches."""
    name = "small/vog_spat"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bcast_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(2):
        r, before, sd = q.get(timeout=300)
        got[r] = (before, sd)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] != got[1][0]                       # they did start apart
    for k in got[0][1]:
        assert np.array_equal(got[0][1][k], got[1][1][k]), k

"""bench.py — queries/sec of the VOGNet forward (gt5, spat, bs=4) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`. For N>1 the driver launches it under
torch.distributed.run (one rank per GPU over RCCL); run plainly with --gpus N > 1 and no
WORLD_SIZE in the environment it re-executes ITSELF under torch.distributed.run on 127.0.0.1.
A "step" = one pass of the
whole hot path over ONE batch of 4 synthetic (query, 4-video) pairs already
resident in HBM: language LSTM -> encoders -> obj_tx -> mul_tx -> score head ->
per-frame arg-max / box gather (-> all-gather of the packed predictions when
N>1). Steps are issued round-robin over `--streams` persistent slots (each a
captured hipGraph with its own inputs, workspace and outputs), so several
batches are in flight — the regime SURVEY.md section 7 calls for, since one batch is
~50 dependent launches. `--streams 1` gives the serial latency.

Before the W warm-up steps the device is warmed up for `device_warmup_ms` (25 ms of the same forwards, untimed, reported in
the JSON line; VOG_BENCH_BURNIN_MS=0 turns it off): the clocks of an idle MI355X are not up after W = 5 steps (0.4 ms), and a
K = 20 region then reads 52 k instead of 55-56 k queries/s. The timed region is exactly K steps between barrier + synchronize.
Side measurements in the same line (never `value`): `value_hbm_inputs` (--rotate-inputs N distinct input sets: features from HBM),
`requests_batched4`, `batch_assembly.measured_host_fed` / `measured_host_fed_graph` (inputs start in pinned host memory every step:
per-call loop / engine.FedPipeline), `training_step`, `cpu_baseline`. The process binds itself to its GPU's NUMA node
(`host_numa_node`; VOG_BENCH_NUMA_BIND=0 leaves the affinity alone).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

ec = importlib.import_module("vognet-pytorch_amd.extended_config")
synth = importlib.import_module("vognet-pytorch_amd.synth")
eng_mod = importlib.import_module("vognet-pytorch_amd.engine")
D = importlib.import_module("vognet-pytorch_amd.dist")

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(mdl="vog", conc="spat", exp="gt5", B=4, tx="bf16",
                 desc="VOGNet spat gt5 (obj_tx+mul_tx, use_rel) bs=4"),
    # not a BASELINE config: four bs=4 requests served as ONE forward (what dynamic batching of the validation loop would run)
    "cfg2x4": dict(mdl="vog", conc="spat", exp="gt5", B=16, tx="bf16",
                   desc="VOGNet spat gt5 (obj_tx+mul_tx, use_rel) 4 x bs=4 in one forward"),
    "cfg2x2": dict(mdl="vog", conc="spat", exp="gt5", B=8, tx="bf16", desc="VOGNet spat gt5, 2 x bs=4 in one forward"),
    "cfg2x8": dict(mdl="vog", conc="spat", exp="gt5", B=32, tx="bf16", desc="VOGNet spat gt5, 8 x bs=4 in one forward"),
    "cfg3": dict(mdl="vog", conc="temp", exp="gt5", B=8, tx="bf16",
                 desc="VOGNet temp gt5 bs=8"),
    "cfg4": dict(mdl="vog", conc="spat", exp="p100", B=4, tx="bf16",
                 desc="VOGNet spat p100 bs=4"),
    "cfg5": dict(mdl="vog", conc="svsq", exp="gt5", B=16, tx="f16",
                 desc="VOGNet svsq gt5 + pred_cmp bs=16 fp16"),
}
VOCAB = 5000
PEAK_MFMA_TFLOPS = 2500.0       # dense bf16/f16, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0           # HBM3E peak (same table; ~6300 GB/s is what streaming kernels achieve)
N_CUS = 256


def make_cfg(w):
    cfg = ec.get_default_cfg()
    ec.update_from_dict(cfg, {"mdl.name": w["mdl"], "ds.conc_type": w["conc"], "ds.exp_setting": w["exp"],
                              "mdl.obj_tx.use_rel": True, "mdl.mul_tx.use_rel": True})
    cfg.hip.tx_dtype = w["tx"]
    return cfg


def kernel_flops(w, T):
    """Algorithmic FLOPs per launch of the hot kernels (dense reference
    formulation, 2*M*N*K) for the gt5/p100 spat/temp/svsq shapes."""
    nppf0 = 5 if w["exp"] == "gt5" else 100
    ncmp = 1 if w["conc"] == "svsq" else 4
    B = w["B"]
    n_vid = B * (ncmp if w["conc"] in ("sep", "svsq") else 1)
    nfrm = ncmp * 10 if w["conc"] == "temp" else 10
    nppf = ncmp * nppf0 if w["conc"] == "spat" else nppf0
    NP = nfrm * nppf
    S_mul, N_mul, d_mul = n_vid * nfrm, 5 * nppf, 768
    S_obj, N_obj, d_obj = n_vid, NP, 512
    rm, ro = S_mul * N_mul, S_obj * N_obj
    f = {
        "mul_qkv": 2.0 * rm * d_mul * 3 * d_mul,          # dense form (layers >= 1 / unstructured input)
        "mul_pv": 2.0 * ro * d_obj * 3 * d_mul,           # structured layer 0: vis part
        "mul_pl": 2.0 * (B * (ncmp if w["conc"] in ("sep", "svsq") else 1) * 5) * 256 * 3 * d_mul,
        "mul_combine": 0.0,
        "mul_attn": 4.0 * S_mul * N_mul * N_mul * d_mul,
        "mul_wo": 2.0 * rm * d_mul * d_mul,
        "mul_ffn1": 2.0 * rm * d_mul * (d_mul // 2),
        "mul_ffn2": 2.0 * rm * d_mul * (d_mul // 2),
        "lin2": 2.0 * rm * d_mul * 256,
        "obj_qkv": 2.0 * ro * d_obj * 3 * d_obj,
        "obj_attn": 4.0 * S_obj * N_obj * N_obj * d_obj,
        "obj_wo": 2.0 * ro * d_obj * d_obj,
        "prop_enc": 2.0 * ro * 2048 * 256,
        "seg_enc": 2.0 * n_vid * (NP // nppf0) * 3072 * 256,
        "lstm_ih0": 2.0 * B * (ncmp if w["conc"] in ("sep", "svsq") else 1) * T * 8192 * 512,
        "lstm_ih1": 2.0 * B * (ncmp if w["conc"] in ("sep", "svsq") else 1) * T * 8192 * 2048,
        "lstm_outproj": 2.0 * B * (ncmp if w["conc"] in ("sep", "svsq") else 1) * (T + 1) * 2048 * 256,
        "obj_ffn1": 2.0 * ro * d_obj * (d_obj // 2), "obj_ffn2": 2.0 * ro * d_obj * (d_obj // 2),
        "obj_tail": 2.0 * ro * d_obj * d_obj + 4.0 * ro * d_obj * (d_obj // 2),
        "mul_tail": 2.0 * rm * d_mul * d_mul + 4.0 * rm * d_mul * (d_mul // 2) + 2.0 * rm * d_mul * 256,
        "vis_enc": 2.0 * ro * 2048 * 256 + 2.0 * n_vid * (NP // nppf0) * 3072 * 256,
        "argvec": 0.0, "enc_finish": 0.0, "cast_feats": 0.0, "mul_ln1": 0.0, "mul_ln2": 0.0,
        "obj_ln1": 0.0, "obj_ln2": 0.0, "score": 0.0, "pred_head": 0.0,
    }
    Bn = B * (ncmp if w["conc"] in ("sep", "svsq") else 1)
    lstm = 2.0 * Bn * T * (2 * 4096 * 512 + 2 * 4096 * 2048 + 4 * 4096 * 1024)
    dense = {k: v for k, v in f.items() if k not in ("mul_pv", "mul_pl", "mul_combine", "lstm_ih0", "lstm_ih1",
                                                     "lstm_outproj", "obj_tail", "mul_tail", "vis_enc")}
    total = sum(dense.values()) + lstm + f["lstm_outproj"]
    # EXECUTED work (structure exploitation is reported against the dense figure, with the
    # reduced figure alongside): layer 0 of mul_tx projects the visual rows and the language rows once each
    # instead of every [vis || lang] token (mul_pv + mul_pl for mul_qkv), and its attention is separable:
    # nppf + nsrl keys per query instead of nsrl * nppf (exact, csrc/attention_dev.h). Everything else runs
    # the dense formulation. (MFMA tile padding - 4 of 16 columns in the BiLSTM, 171 of 192 head columns in
    # obj_tx - is NOT counted: it is work the hardware does, not work the algorithm needs.)
    executed = dict(dense)
    if w["mdl"] == "vog":
        executed["mul_qkv"] = f["mul_pv"] + f["mul_pl"]
        executed["mul_attn"] = f["mul_attn"] * (nppf + 5) / float(N_mul)
        if nppf > 32:
            # several visual key blocks: the E x F form (csrc/attn_struct_ef_dev.h) also shares Q.K^T among the 5 arguments of
            # a proposal (half of the attention FLOPs are Q.K^T: 1/5 of that half; P.V keeps its size)
            executed["mul_attn"] *= 0.5 * (1.0 / 5.0) + 0.5
    total_exec = sum(executed.values()) + lstm + f["lstm_outproj"]
    f["_executed_total"] = total_exec
    f["_executed_mul_attn"] = executed.get("mul_attn", 0.0)
    return f, total


def lstm_step_bytes(w, T):
    """Algorithmic HBM bytes of ONE recurrent step launch (both directions of one layer):
    W_hh is streamed once per step (2 dirs x 4R x R 16-bit: it cannot stay on chip between
    dependent launches), plus the step's slice of the input-projection gates, the fp32 cell
    state (read + write) and the 16-bit hidden state in / out."""
    R = 1024
    ncmp = 1 if w["conc"] == "svsq" else 4
    Bn = w["B"] * (ncmp if w["conc"] in ("sep", "svsq") else 1)
    whh = 2 * 4 * R * R * 2
    gates = 2 * Bn * 4 * R * 4
    state = 2 * Bn * R * (4 + 4 + 2 + 2 + 2)
    return whh + gates + state


def pmc_traffic(workload):
    """HBM traffic measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950
    correction applied), committed under profiles/ by scratch/run_round_profiles.sh."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except Exception:
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; logical count // 2 if unreadable."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


_ORIG_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None     # before any NUMA binding


def _set_affinity_all_threads(mask):
    """sched_setaffinity on EVERY thread of the process (torch's intra-op pool threads keep the mask they were created with)."""
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), mask)
            except OSError:
                pass
    except Exception:
        pass


def cpu_baseline(w, cfg, sd, batch, budget_s=14.0):
    """The CPU oracle (a validated port of the reference forward, oracle/vog_oracle.py: torch-CPU fp32
    ops, manual LSTM loops; "kind": "port" - the reference itself measured 43 queries/s on cfg 2 in the
    survey probe, this port 41-42) timed on this host on the same workload, as SURVEY.md 8(d) asks:
    8 / 16 / 32 threads AND every CPU this process is ALLOWED to run on (sched_getaffinity as the process started - the
    NUMA binding of the GPU measurement is lifted for this leg and restored after it; round 4 timed 128 threads inside a
    one-node mask: 0.22 queries/s), CPU model stated. `value` is the best of them and `cores` the thread count it used (the
    forward is a chain of small GEMMs, M = 4..4000: more threads is not faster)."""
    from oracle import vog_oracle as vo
    bound = os.sched_getaffinity(0) if _ORIG_AFFINITY is not None else None
    if _ORIG_AFFINITY is not None:
        _set_affinity_all_threads(_ORIG_AFFINITY)
    try:
        return _cpu_baseline(vo, w, cfg, sd, batch, budget_s)
    finally:
        if bound is not None:
            _set_affinity_all_threads(bound)


def _cpu_baseline(vo, w, cfg, sd, batch, budget_s):
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nphys = min(physical_cores(), ncpu)
    oc = vo.OracleCfg.from_cfg(cfg, VOCAB, ec.num_prop_per_frm(cfg))
    sdt, inp = vo.to_torch(sd), vo.to_torch(batch)
    tried = {}
    cands = sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), nphys})
    WARM, TIMED = 5, 20                    # SURVEY.md 8(d) / BASELINE.md 3: >= 5 warm-up + >= 20 timed batches, median
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            t_start = time.time()
            share = budget_s / len(cands)
            nwarm = 0
            while nwarm < WARM and (nwarm == 0 or time.time() - t_start < share / 4):
                vo.pred_head(oc, vo.forward(oc, sdt, inp), inp)
                nwarm += 1
            times = []
            while len(times) < TIMED and (not times or time.time() - t_start < share):
                t0 = time.perf_counter()
                vo.pred_head(oc, vo.forward(oc, sdt, inp), inp)
                times.append(time.perf_counter() - t0)
            tried[nt] = (float(np.median(times)), len(times), nwarm)
    if not tried:
        return {"value": None, "unit": "queries/s", "cores": ncpu, "kind": "port",
                "sample": "no forward finished inside the time budget"}
    best = min(tried, key=lambda k: tried[k][0])
    med, n, nwarm = tried[best]
    return {"value": w["B"] / med, "unit": "queries/s", "cores": best, "kind": "port",
            "cpu_model": cpu_model(), "physical_cores": nphys, "logical_cpus": ncpu, "allowed_cpus": ncpu,
            "by_threads": {str(k): w["B"] / v[0] for k, v in tried.items()},
            "warmup_forwards": nwarm, "timed_forwards": n,
            "sample": f"{n} timed forwards of the same batch (bs={w['B']}) after {nwarm} warm-up, median"
                      + ("" if (n >= TIMED and nwarm >= WARM) else f" (the {budget_s:.0f} s budget ended the sample early: "
                         f"protocol is {WARM} + {TIMED})") + "; "
                      f"threads tried: " + ", ".join(f"{k}: {w['B'] / v[0]:.1f} q/s" for k, v in tried.items()),
            "ms_per_batch": med * 1e3}


def check_parity(w, cfg, sd, batch, slot, eng):
    """slot.out (device) vs the CPU oracle (oracle/vog_oracle.py, pinned against the reference) on the same
    batch: max relative error of mdl_outs_eval / pred_scores where the reference is non-zero (bound 1e-3,
    north_star), masked entries exactly zero, logits within 6e-3 abs."""
    from oracle import vog_oracle as vo
    nppf0 = ec.num_prop_per_frm(cfg)
    oc = vo.OracleCfg.from_cfg(cfg, VOCAB, nppf0)
    inp = vo.to_torch(batch)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        ref = vo.forward(oc, vo.to_torch(sd), inp)
        rp = vo.pred_head(oc, ref, inp)
    ev, rev = slot.out["mdl_outs_eval"].float().cpu(), ref["mdl_outs_eval"]
    nz = rev != 0
    rel = float(((ev - rev).abs() / rev.abs().clamp(min=1e-6))[nz].max()) if bool(nz.any()) else 0.0
    masked_zero = bool((ev[~nz] == 0).all())
    logit = float((slot.out["mdl_outs"].float().cpu() - ref["mdl_outs"]).abs().max())
    ncmp = batch["new_srl_idxs"].shape[1]
    pred = eng.unpack_pred(slot.out["pred_rec"], ncmp)
    sc, rsc = pred["scores"].float().cpu(), rp["scores"]
    snz = rsc != 0
    srel = float(((sc - rsc).abs() / rsc.abs().clamp(min=1e-6))[snz].max()) if bool(snz.any()) else 0.0
    # pred_boxes: an arg-max gather - equal to the oracle's except where its top proposals of a frame are a near tie; the number of
    # such flips is part of the record (bound: 0.5 % of the boxes, each flipped box's score within 2e-3 of the oracle's maximum)
    bx, rbx = pred["boxes"].float().cpu(), rp["boxes"]
    flip = (bx != rbx).any(-1)
    nflip, nbox = int(flip.sum()), int(flip.numel())
    flip_ok = nflip <= 0.005 * nbox and bool((((sc - rsc).abs() / rsc.abs().clamp(min=1e-6))[flip] <= 2e-3).all())
    ok = bool(np.isfinite(rel) and np.isfinite(srel) and rel <= 1e-3 and srel <= 1e-3 and logit <= 6e-3 and masked_zero and flip_ok)
    return {"ok": ok, "rel_err_mdl_outs_eval": rel, "rel_err_pred_scores": srel, "abs_err_logits": logit,
            "box_flips": nflip, "boxes": nbox, "box_flips_bound": int(0.005 * nbox),
            "masked_entries_exactly_zero": masked_zero, "bound_rel": 1e-3, "bound_logit_abs": 6e-3,
            "against": "CPU oracle (oracle/vog_oracle.py) on slot 0's batch, outputs of its last timed launch"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run (one rank
    per GPU, rendezvous on 127.0.0.1) and relay its exit code; rank 0 of the child prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cobatch", type=int, default=1,
                    help="G > 1: the language encoder (BiLSTM) of G in-flight batches runs as one pass "
                         "(W_hh streamed once per recurrent step for all of them); every batch keeps its own "
                         "inputs/outputs and its stand-alone results (tests/test_gpu_forward.py)")
    ap.add_argument("--lstm-steps", action="store_true",
                    help="BiLSTM as 2T step launches (lowest single-batch latency; re-streams W_hh every step) "
                         "instead of the default one persistent launch per layer (W_hh resident in registers, "
                         "h handed between workgroups through memory; higher throughput with batches in flight)")
    ap.add_argument("--no-cobatch-extra", action="store_true",
                    help="skip the second timed run that reports the co-batched language encoder (G=4) "
                         "beside the strict per-batch figure")
    ap.add_argument("--cobatch-extra", action="store_true",
                    help="also time round 2's shared language encoder (4 in-flight batches per BiLSTM pass): `lang_cobatch4`")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--set", action="append", metavar="OPTION=INT", help="vog_ctx_set_int switch (A/B measurements)")
    ap.add_argument("--no-train-extra", action="store_true", help="skip the side measurement of the fp32 training step")
    ap.add_argument("--kernel-iters", type=int, default=100)
    ap.add_argument("--rotate-inputs", type=int, default=64,
                    help="N > 0: also time the strict path with N distinct device-resident input sets cycling through the "
                         "streams' workspaces (N x 8.7 MB of features > the 256 MB Infinity Cache at cfg 2), reported as "
                         "`value_hbm_inputs` beside `value`; 0 = skip")
    ap.add_argument("--rotate-main", action="store_true",
                    help="apply --rotate-inputs to the main timed region too (A/B experiments with --throughput-only)")
    ap.add_argument("--throughput-only", action="store_true",
                    help="print '<queries/s> <us/step>' and exit (ablation experiments, scratch/ablate.sh)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_dist.py: two ranks on the ONE GPU of the test box): device index and collective backend
    if "VOG_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["VOG_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    # VOG_BENCH_FORCE_DIST=1: run the N > 1 exchange path (RCCL all-gather per step) with one rank too
    use_dist = world > 1 or bool(os.environ.get("VOG_BENCH_FORCE_DIST"))
    out = sys.stdout
    if use_dist:
        # RCCL prints a version banner through C stdio (it surfaced AFTER the JSON line at exit): keep the
        # process's stdout for the one JSON line, send everything else written to fd 1 to stderr
        sys.stdout.flush()
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        if "RANK" not in os.environ:                     # forced single-rank group: fill in the rendezvous
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=os.environ.get("VOG_BENCH_BACKEND", "nccl"), init_method="env://")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        # one stream per lane of the engine's book (4; with a multi-rank group the last lane yields to a pending gather:
        # engine._lane_enter / dist.RecordRing - between collectives all 4 forwards are in flight)
        args.streams = min(args.streams, eng_mod._max_inflight())
    dev = torch.device("cuda", local_rank)
    # the rank's host side on its GPU's NUMA node (a launcher's numactl; VOG_BENCH_NUMA_BIND=0: leave the affinity alone)
    numa_node = D.bind_host_to_device_node(local_rank) if os.environ.get("VOG_BENCH_NUMA_BIND", "1") == "1" else None

    w = WORKLOADS[args.workload]
    cfg = make_cfg(w)
    nppf0 = ec.num_prop_per_frm(cfg)
    comm = {"vocab_size": VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1},
            "num_prop_per_frm": nppf0}
    sd = synth.init_state_dict(cfg, VOCAB, seed=1)
    eng = eng_mod.VogEngine(cfg, comm)
    eng.load_state_dict(sd)
    # the persistent layer kernel needs all its 64 workgroups co-resident: safe up to 4 of them in
    # flight (4 hardware queues x 64 workgroups = 256 CUs); a stalled hand-off poisons the output with NaN
    persistent = not args.lstm_steps
    eng.set_option("lstm_persistent", int(persistent))
    for kv in args.set or []:            # engine switches for A/B runs, e.g. --set pair_launches=0 (defaults are what `value` is for)
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    cfg_id = int(args.workload[3:4])

    # ONE set of streams for every measurement of this process. Which hardware queue a HIP stream lands on depends on the
    # order in which ALL streams of the process were created and first used (scratch/dbg_streams2.py: the same 4-stream loop
    # runs at 71 or 110 us per batch depending on whether the slots' capture streams were created between them; streams i and
    # i + 4 of a run share a queue): the streams are created here, back to back, before any slot exists, and reused. (Picking
    # them by a pairwise timing test was tried: what overlaps during the test does not reliably overlap afterwards.)
    stream_pool = [torch.cuda.Stream(device=dev) for _ in range(16)]
    if os.environ.get("VOG_PERF_EXPERIMENTS") and os.environ.get("VOG_BENCH_STREAM_IDS"):
        # experiment only: which members of the pool carry the forwards (e.g. "0,2,5,7")
        ids = [int(x) for x in os.environ["VOG_BENCH_STREAM_IDS"].split(",")]
        stream_pool = [stream_pool[i] for i in ids] + [st for i, st in enumerate(stream_pool) if i not in ids]
    if os.environ.get("VOG_PERF_EXPERIMENTS") and os.environ.get("VOG_BENCH_CU_MASK"):
        # experiment only (no `value` is printed under VOG_PERF_EXPERIMENTS): streams restricted to a subset of the CUs
        # (hipExtStreamCreateWithCUMask) - "is the 4-stream loop bound by CU time?" VOG_BENCH_CU_MASK = a 32-bit hex pattern
        # repeated over the CU bit array (e.g. 77777777 = 3 of every 4 CUs), or per-stream patterns separated by ','
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        pats = [int(x, 16) for x in os.environ["VOG_BENCH_CU_MASK"].split(",")]
        stream_pool = []
        for i in range(16):
            words = (ctypes.c_uint32 * 8)(*([pats[i % len(pats)]] * 8))
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
            assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
            stream_pool.append(torch.cuda.ExternalStream(h.value, device=dev))

    host_issue = [0.0]

    def measure(G, steps, warmup, batched=False, rotate=0, eng=eng):
        """K timed steps (one step = one batch of the workload) after W warm-up steps with G
        batches per language-encoder pass. Returns (seconds, slots, batches, in-flight count).
        rotate = R > 0 (graph mode, G = 1): R distinct device-resident input sets per stream are cycled through the
        stream's workspace (slot j runs on stream j % streams and shares that stream's workspace), so that a forward's
        features come from HBM, not from the Infinity Cache a 4-slot replay loop leaves them in."""
        nunits = max(1, args.streams)                                 # units in flight (slot or group)
        nsets = max(1, rotate) if G == 1 else 1
        nstreams = nunits * G * nsets                                 # batch buffers (in flight: nunits * G)
        slots, streams, batches, units = [], [], [], []
        for s in range(nstreams):
            b = synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=VOCAB,
                                 seed=1000 * cfg_id + rank * 64 + s)
            batches.append(b)
            su = s % nunits if nsets > 1 else s
            streams.append(stream_pool[su] if su < len(stream_pool) else torch.cuda.Stream(device=dev))
        # the prediction records of all batches in flight live in ONE buffer: the exchange step is
        # one all-gather per round of in-flight batches instead of one per batch
        ncmp_w = 1 if w["conc"] == "svsq" else 4
        recbuf = torch.zeros(nstreams * w["B"], eng.record_words(ncmp_w), dtype=torch.float32, device=dev)
        if G == 1:
            for s_, b in enumerate(batches):
                slots.append(eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()},
                                           graph=not args.no_graph,
                                           pred_rec=recbuf[s_ * w["B"]:(s_ + 1) * w["B"]],
                                           share_ws_with=slots[s_ % nunits] if (nsets > 1 and s_ >= nunits) else None))
            units = slots
        else:
            assert steps % G == 0 and warmup % G == 0, "--steps / --warmup must be multiples of the co-batch size"
            for u in range(nunits * nsets):
                mk = eng.make_batched if batched else eng.make_group
                grp = mk([{k: torch.from_numpy(v) for k, v in b.items()}
                          for b in batches[u * G:(u + 1) * G]], graph=not args.no_graph,
                         pred_rec=recbuf[u * G * w["B"]:(u + 1) * G * w["B"]])
                units.append(grp)
                slots.extend(grp.slots)
        # ---- exchange step (N > 1): the packed prediction records of every batch are copied (27 KB,
        # on the batch's own stream) into a ring of RING batches; ONE RCCL all-gather moves half a ring
        # at a time. A collective per batch was measured to cost 12-20 us per step on one GPU already:
        # it runs on the process group's own (5th) stream, which shares a hardware queue with one of
        # the four forward streams and drags that stream's chain behind the collective's dependencies.
        unit_rows = G * w["B"]
        per_half = max(nunits, (64 // nunits) * nunits)           # unit launches per half ring (64: the last lane waits for a gather once per half)
        # (VOG_FORCE_MULTI_RANK_LANES=1 with one rank: the half-ring copy stands in for the collective, so that the event-gated
        # fourth lane of engine._lane_enter is exercised and priced on one GPU)
        forced_lanes = os.environ.get("VOG_FORCE_MULTI_RANK_LANES") == "1"
        ring = D.RecordRing(unit_rows, recbuf.shape[1], per_half, dev,
                            on_half=(lambda g_, n_: None) if forced_lanes else None) if use_dist else None

        def step(i):                       # graph mode: one unit (slot, or group of G batches) per call
            u = i % (nunits * nsets)       # (rotation: unit u lives on stream u % nunits, see above)
            units[u].launch(streams[u])
            if ring is not None:
                ring.push(recbuf[u * unit_rows:(u + 1) * unit_rows], streams[u])

        def fence():
            if ring is not None:
                ring.flush()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        def run(nsteps):
            for i in range(nsteps // G):
                step(i)

        # device warm-up in front of the W warm-up steps (untimed, reported as `device_warmup_ms`): an idle MI355X needs some
        # milliseconds of work before its clocks are up - 5 warm-up steps are 0.4 ms. K = 20 from a cold device: 52.2 k queries/s,
        # after 20 ms of forwards: 55.3-56.4 k; K = 400 does not care (scratch/r4_burn.sh). VOG_BENCH_BURNIN_MS=0 turns it off.
        burn_ms = float(os.environ.get("VOG_BENCH_BURNIN_MS", "25"))
        if burn_ms > 0:
            tb = time.perf_counter()
            while True:
                run(4 * max(1, nunits) * G)
                torch.cuda.synchronize()
                stop = (time.perf_counter() - tb) * 1e3 >= burn_ms
                if use_dist:                     # every rank runs the same number of rounds (the exchange ring's collectives must pair up)
                    t = torch.tensor([1 if stop else 0], device=dev, dtype=torch.int32)
                    dist.broadcast(t, src=0)
                    stop = bool(int(t.item()))
                if stop:
                    break
        run(warmup)
        fence()
        t0 = time.perf_counter()
        run(steps)
        host_issue[0] = time.perf_counter() - t0      # the host's share: issuing `steps` launches (the device runs behind it)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, slots, batches, nunits * G

    G = max(1, args.cobatch)
    rot_sets = (args.rotate_inputs + max(1, args.streams) - 1) // max(1, args.streams) if args.rotate_inputs > 0 else 0
    dt, slots, batches, nstreams = measure(G, args.steps, args.warmup, rotate=rot_sets if args.rotate_main else 0)
    host_issue_main = host_issue[0]
    T = slots[0].T
    # a short timed region (the driver's K = 20) carries a fixed ~160 us of pipeline fill and drain (4 forwards in flight,
    # a forward takes ~280 us under load): the same strict path timed over 400 steps is reported BESIDE `value`
    steady = None
    if G == 1 and world == 1 and args.steps < 200 and not args.throughput_only:
        dts, sl_s, _, n_s = measure(G, 400, 40)
        steady = {"value": 400 * w["B"] / dts, "unit": "queries/s", "ms_per_step": dts / 400 * 1e3, "steps": 400, "warmup": 40,
                  "batches_in_flight": n_s,
                  "what": "the same strict per-batch path over 400 timed steps: `value` above is K = %d steps, whose fixed pipeline "
                          "fill / drain (~160 us per timed region) is %.0f %% of its time" % (args.steps, 100 * (1 - (dts / 400) / (dt / args.steps)))}
        del sl_s
    # the same strict path with f16 transformers (`cfg.hip.tx_dtype = auto`, the package default since round 5: same MFMA rate,
    # three more mantissa bits - it holds the 1e-3 bound to wq / wk x 12 where bf16 leaves it at x 8)
    f16_tx = None
    if G == 1 and world == 1 and w["tx"] == "bf16" and not args.throughput_only and not args.no_graph:
        cfg_h = make_cfg(dict(w, tx="auto"))
        eng_h = eng_mod.VogEngine(cfg_h, comm)
        eng_h.load_state_dict(sd)
        eng_h.set_option("lstm_persistent", int(persistent))
        for kv in args.set or []:
            k_, v_ = kv.split("=")
            eng_h.set_option(k_, int(v_))
        assert eng_h.precise is None and eng_h.desc.tx_dtype == eng_mod.L.VOG_F16
        dth_, sl_h16, _, n_h16 = measure(G, 400, 40, eng=eng_h)
        par_h = check_parity(w, cfg_h, sd, batches[0], sl_h16[0], eng_h)
        ref_ms = (steady["ms_per_step"] if steady is not None else dt / args.steps * 1e3)
        f16_tx = {"value": 400 * w["B"] / dth_ if par_h["ok"] else None, "unit": "queries/s", "ms_per_step": dth_ / 400 * 1e3,
                  "steps": 400, "warmup": 40, "batches_in_flight": n_h16, "attention_sharpness": eng_h.sharpness,
                  "ratio_vs_bf16_same_protocol": ref_ms / (dth_ / 400 * 1e3) if steady is not None or args.steps >= 200 else None,
                  "parity": {k_: par_h[k_] for k_ in ("ok", "rel_err_mdl_outs_eval", "rel_err_pred_scores", "abs_err_logits")},
                  "what": "the same strict per-batch path, 400 timed steps, f16 operands in obj_tx / mul_tx (the package default; "
                          "`value` is bf16 as BASELINE.json's config names it)"}
        del sl_h16, eng_h
    # the same strict path for a checkpoint whose attention is too sharp for plain 16-bit logits (wq / wk x 16: logit std ~6 / ~30
    # nats in obj_tx / mul_tx): `auto` plans hi + lo f16 operands for it (three MFMAs for everything that feeds attention logits;
    # engine.py) - round 5 ran such checkpoints on the fp32 path at 1.7 k queries/s
    hi_lo = None
    if G == 1 and world == 1 and cfg_id in (2, 3, 5) and not args.throughput_only and not args.no_graph:
        try:
            sd_s = synth.sharpen_state_dict({k: np.array(v, copy=True) for k, v in sd.items()}, 16.0, 4.0)
            cfg_s = make_cfg(dict(w, tx="auto"))
            eng_s = eng_mod.VogEngine(cfg_s, comm)
            eng_s.load_state_dict(sd_s)
            eng_s.set_option("lstm_persistent", int(persistent))
            dts_, sl_s2, b_s2, n_s2 = measure(G, 400, 40, eng=eng_s)
            par_s = check_parity(w, cfg_s, sd_s, b_s2[0], sl_s2[0], eng_s)
            lo_, lm_ = eng_s.observed_logit_max()
            hi_lo = {"value": 400 * w["B"] / dts_ if (par_s["ok"] and eng_s.plan == "split") else None, "unit": "queries/s",
                     "ms_per_step": dts_ / 400 * 1e3, "steps": 400, "warmup": 40, "batches_in_flight": n_s2, "plan": eng_s.plan,
                     "attention_sharpness": eng_s.sharpness, "observed_logit_max_nats": {"obj_tx": lo_, "mul_tx": lm_},
                     "parity": {k_: par_s[k_] for k_ in ("ok", "rel_err_mdl_outs_eval", "rel_err_pred_scores", "abs_err_logits", "box_flips")},
                     "what": "the same strict per-batch path with wq / wk of both transformers x 16 (pe x 4): outside the f16 envelope, "
                             "`auto` runs the hi + lo operand kernels; parity against the CPU oracle with the same weights"}
            del sl_s2, eng_s
        except Exception as e:          # never fail the bench line on the side measurement
            hi_lo = {"value": None, "error": str(e)}
    # the same strict path with the inputs coming from HBM: N distinct input sets cycle through the streams' workspaces
    hbm_inputs = None
    if G == 1 and world == 1 and rot_sets > 1 and not args.no_graph and not args.throughput_only and not args.rotate_main:
        ksteps = max(400, args.steps)
        dth, sl_h, _, n_h = measure(G, ksteps, max(40, rot_sets * max(1, args.streams)), rotate=rot_sets)
        nfin_h = int(sum(int((~torch.isfinite(sl.out["mdl_outs_eval"])).sum().item()) for sl in sl_h))
        in_bytes = sum(v.numel() * v.element_size() for k, v in sl_h[0].inp.items())
        hbm_inputs = {"value": ksteps * w["B"] / dth if nfin_h == 0 else None, "unit": "queries/s", "ms_per_step": dth / ksteps * 1e3,
                      "steps": ksteps, "input_sets": len(sl_h), "input_bytes_per_set": in_bytes,
                      "input_bytes_total": in_bytes * len(sl_h), "batches_in_flight": n_h,
                      "what": "the strict per-batch path with %d distinct device-resident input sets (%.0f MB in total, the Infinity "
                              "Cache holds 256 MB) cycling through the %d streams' workspaces: every forward reads its features "
                              "from HBM; `value` replays %d slots whose 35 MB of features stay cache-resident"
                              % (len(sl_h), in_bytes * len(sl_h) / 1e6, n_h, n_h)}
        del sl_h
    # four bs=4 requests served as ONE forward (dynamic batching; `make_batched`): reported beside `value`, never instead of
    # it - `value` is one bs=4 forward per launch sequence
    batched4 = None
    if G == 1 and world == 1 and not args.throughput_only and not args.no_graph and not args.no_cobatch_extra:
        dtb, sl_b, _, n_b = measure(4, 400, 40, batched=True)
        nfin = int(sum(int((~torch.isfinite(sl.out["mdl_outs_eval"])).sum().item()) for sl in sl_b))
        batched4 = {"value": 400 * w["B"] / dtb if nfin == 0 else None, "unit": "queries/s", "ms_per_step": dtb / 400 * 1e3, "steps": 400,
                    "warmup": 40, "requests_per_forward": 4, "forwards_in_flight": n_b // 4,
                    "what": "400 timed steps (one step = one bs=%d request); four requests share one forward: every kernel of the chain "
                            "is four times wider, the BiLSTM fills its 16 MFMA columns, a request costs a quarter of the launches; each "
                            "request's outputs equal its stand-alone forward (tests/test_gpu_forward.py::"
                            "test_batched_requests_match_standalone_forwards); Evaluator.forward uses it with cfg.hip.batch_requests" % w["B"]}
        del sl_b
    # second, separately timed run of the same K steps: the language encoder of 4 in-flight batches as
    # one pass (reported beside `value`, never instead of it: `value` is the strict per-batch path)
    extra = None
    if not args.cobatch_extra:
        extra = {"value": None, "skipped": "opt-in since round 5 (--cobatch-extra): round 2's shared language encoder is slower "
                                           "than four requests per forward (`requests_batched4`), which replaced it"}
    elif G == 1 and not args.no_cobatch_extra and not args.throughput_only and not args.no_graph:
        # (K and W rounded up to multiples of the group size: the driver's --warmup 5 used to drop this block silently)
        k4, w4 = (args.steps + 3) // 4 * 4, (args.warmup + 3) // 4 * 4
        dt4, slots4, _, n4 = measure(4, k4, w4)
        assert np.isfinite(float(slots4[0].out["mdl_outs_eval"].sum().item()))
        extra = {"value": world * k4 * w["B"] / dt4, "unit": "queries/s", "ms_per_step": dt4 / k4 * 1e3,
                 "batches_in_flight": n4, "steps": k4, "warmup": w4,
                 "what": "same K steps, same per-batch inputs/outputs; the BiLSTM language encoder of 4 in-flight "
                         "batches runs as one pass (W_hh streamed once per recurrent step for the 16 sentences "
                         "instead of once per batch), transformers/heads per batch as before; every batch's "
                         "outputs match its stand-alone forward (tests/test_gpu_forward.py::test_group_*)"}
        del slots4
    if args.throughput_only:
        if rank == 0:
            print(f"{world * args.steps * w['B'] / dt:.1f} {dt / args.steps * 1e6:.2f} host_issue_us_per_step {host_issue_main / args.steps * 1e6:.2f}", file=out, flush=True)
        if use_dist:
            dist.destroy_process_group()
        return
    # every slot's outputs must be finite (a timed-out BiLSTM hand-off poisons its forward with NaN)
    nan_count = int(sum(int((~torch.isfinite(sl.out["mdl_outs_eval"])).sum().item()) +
                        int((~torch.isfinite(sl.out["mdl_outs"])).sum().item()) for sl in slots))
    if use_dist:
        t = torch.tensor([nan_count], device=dev, dtype=torch.int64)
        dist.all_reduce(t)
        nan_count = int(t.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    value = world * args.steps * w["B"] / dt
    # what was timed is checked: slot 0's outputs (its last timed launch) against the CPU oracle on the
    # same batch, with the tolerances of tests/test_gpu_forward.py
    parity = check_parity(w, cfg, sd, batches[0], slots[0], eng)
    parity["non_finite_outputs_all_slots"] = nan_count
    parity["ok"] = bool(parity["ok"] and nan_count == 0)
    experiments = bool(os.environ.get("VOG_PERF_EXPERIMENTS"))
    res = {
        "metric": "queries/sec (VOGNet forward, gt5 spat, bs=4)",
        # no number without a result: experiment knobs (VOG_PERF_EXPERIMENTS: skipped steps, forced tiles)
        # or a failed parity check make the timing meaningless
        "value": value if (parity["ok"] and not experiments) else None, "unit": "queries/s",
        "parity": parity,
        "n_gpus": world, "host_numa_node": numa_node, "device_warmup_ms": float(os.environ.get("VOG_BENCH_BURNIN_MS", "25")), "rccl_ranks": dist.get_world_size() if use_dist else 1, "per_rank_value": value / world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "host_issue_ms_per_step": host_issue_main / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": w["tx"], "data": "synthetic",
        "config": {"workload": w["desc"], "global_batch": world * w["B"], "batch_per_gpu": w["B"],
                   "sentence_len": T, "batches_in_flight": nstreams, "lang_cobatch": G,
                   "lstm": "persistent layer kernel (1 launch per layer)" if persistent else "step launches (2T per forward)",
                   "submission": f"hipGraph on {nstreams} HIP streams",
                   "weights": "seeded default-init-like, vocab 5000",
                   "launches_per_forward": "see kernels_usec: steps joined by '+' share one launch (csrc/pair.hip)",
                   "parallelism": f"dp{world} (replicated weights; prediction records of every batch staged in a "
                                  f"device ring, one RCCL all-gather per 64 batches)"},
    }
    if extra:
        res["lang_cobatch4"] = extra
    if steady is not None:
        res["steady_state_400_steps"] = steady
    if f16_tx is not None:
        res["f16_transformers"] = f16_tx
    if hi_lo is not None:
        res["hi_lo_plan_sharp16"] = hi_lo
    if hbm_inputs is not None:
        res["value_hbm_inputs"] = hbm_inputs["value"]
        res["hbm_inputs"] = hbm_inputs
    if batched4 is not None:
        res["requests_batched4"] = batched4
    # ---- roofline of the dominant kernel: HIP-event timing inside libvog_hip on this stream
    flops, total_flops = kernel_flops(w, T)
    executed_total = flops.pop("_executed_total")
    executed_mul_attn = flops.pop("_executed_mul_attn")
    ktimes = {}
    names = list(flops) + ["prep", "lstm_layer#0", "lstm_layer#1", "lstm_layer+vis_enc", "lstm_layer+obj_tail",
                           "lstm_outproj+mul_pv", "lstm_step"]
    for k in names:
        try:
            ktimes[k] = eng.time_kernel(slots[0], k, args.kernel_iters)
        except Exception:           # a step name absent for this model variant / option set
            ktimes[k] = None
    lstm_us = None
    for nm in ("lstm_layer#0", "lstm_step"):
        if ktimes.get(nm):
            lstm_us = (nm.split("#")[0], ktimes[nm])
            break
    # dominant kernel = largest share of a forward's kernel time (launches x duration). At the gt5
    # shapes that is the persistent BiLSTM layer (2 launches; latency bound, its roofline is HBM: W_hh
    # streams once per launch); the largest MFMA kernel is the fused mul_tx tail.
    pmc = pmc_traffic(args.workload + ("" if persistent else "_lstm_steps")) or {}
    cands = [k for k in flops if ktimes.get(k) and flops[k] > 0]
    mfma_dom = max(cands, key=lambda k: ktimes[k])
    # FLOPs of the dominant kernel: what it EXECUTES where that is less than the dense formulation (mul_attn)
    fl_dom = executed_mul_attn if mfma_dom == "mul_attn" else flops[mfma_dom]
    ach = fl_dom / (ktimes[mfma_dom] * 1e-6) / 1e12
    # row-block kernels occupy ceil(rows / 64) CUs, not the chip: also quote the fraction of THEIR CUs' peak
    roof_mfma = {"bound": "mfma", "kernel": mfma_dom, "achieved": ach, "peak": PEAK_MFMA_TFLOPS,
                 "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_TFLOPS,
                 "traffic": (pmc.get("kernels", {}).get(mfma_dom) or {}).get("bytes_per_launch"),
                 "usec_per_launch": ktimes[mfma_dom], "flops_per_launch": fl_dom,
                 "launches_per_forward": 1}
    if mfma_dom == "mul_attn":
        roof_mfma["frac_dense_equivalent"] = flops[mfma_dom] / (ktimes[mfma_dom] * 1e-6) / 1e12 / PEAK_MFMA_TFLOPS
        roof_mfma["note"] = "executed FLOPs (separable attention); the dense-equivalent fraction is quoted beside it only"
    if mfma_dom in ("mul_tail", "obj_tail"):
        nppf0 = 5 if w["exp"] == "gt5" else 100
        ncmp = 1 if w["conc"] == "svsq" else 4
        rows = w["B"] * ncmp * 10 * nppf0 * (5 if mfma_dom == "mul_tail" else 1)   # proposals (x nsrl argument slots)
        wgs = (rows + 63) // 64
        act = min(N_CUS, wgs)
        roof_mfma["workgroups"] = wgs
        roof_mfma["active_cus"] = act
        roof_mfma["frac_of_active_cus_peak"] = ach / (PEAK_MFMA_TFLOPS * act / N_CUS)
        roof_mfma["note"] = ("one 64-row block per workgroup, one workgroup per CU: the launch uses `active_cus` of "
                             "the 256 CUs (the others serve the batches in flight on the other streams)")
    res["roofline"] = roof_mfma
    if lstm_us and lstm_us[0] == "lstm_layer" and 2 * lstm_us[1] > ktimes[mfma_dom]:
        # persistent layer kernel: W_hh is read ONCE per launch and kept in registers for all T steps
        nbytes = lstm_step_bytes(w, T) + (T - 1) * (lstm_step_bytes(w, T) - 2 * 4 * 1024 * 1024 * 2)
        us_launch = lstm_us[1]
        # what each layer launch reads besides W_hh and the states: its input projection where the kernel computes it in its
        # prologue (W_ih once per launch - 2 dirs x 4R x K 16-bit, K = 512 for layer 0, 2R for layer 1 - plus the 16-bit layer
        # input), or (round 6, layer 0 beyond 80 columns) the gate-table rows of the batch's tokens; a projection that runs as
        # a GEMM launch (lstm_ih*) is that launch's traffic, not the layer's. Figures are the average of the two launches.
        R, E = 1024, 512
        Bn = w["B"] * ((1 if w["conc"] == "svsq" else 4) if w["conc"] in ("sep", "svsq") else 1)
        both = bool(ktimes.get("lstm_layer#0") and ktimes.get("lstm_layer#1"))
        proj = ["gemm launch", "gemm launch"]
        extra = 0
        if both and not ktimes.get("lstm_ih0"):
            if Bn * T <= 80:
                proj[0] = "kernel prologue"; extra += 2 * 4 * R * E * 2 + Bn * T * E * 2
            else:
                proj[0] = "gate table"; extra += Bn * T * 2 * 4 * R * 4
        if both and not ktimes.get("lstm_ih1"):
            proj[1] = "kernel prologue"; extra += 2 * 4 * R * 2 * R * 2 + Bn * T * 2 * R * 2
        fused_ih = proj[0] != "gemm launch" or proj[1] != "gemm launch"
        if both:
            nbytes = nbytes + extra // 2
            us_launch = 0.5 * (ktimes["lstm_layer#0"] + ktimes["lstm_layer#1"])
        ach_b = nbytes / (us_launch * 1e-6) / 1e9
        res["roofline"] = {"bound": "latency (cache stream)", "governing_peak": "hbm", "kernel": "lstm_layer", "achieved": ach_b, "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": ach_b / PEAK_HBM_GBS,
                           "traffic": (pmc.get("kernels", {}).get("lstm_layer") or {}).get("bytes_per_launch"),
                           "usec_per_launch": us_launch, "bytes_per_launch": nbytes, "launches_per_forward": 2,
                           "input_projection_in_kernel": bool(fused_ih), "input_projection": {"layer0": proj[0], "layer1": proj[1]},
                           "bytes_source": "Infinity Cache (MALL), not HBM: the 88 MB of 16-bit weights stay resident in the 256 MB "
                                           "cache between launches (the PMC `traffic` is the L2's memory-side request counter, which "
                                           "counts cache hits too); `peak` is the HBM figure the bench contract names - read `frac` as "
                                           "a weight-stream rate against the HBM peak, not as HBM utilisation",
                           "share_of_forward_kernel_time": 2 * us_launch / max(1e-9, sum(
                               ktimes[k] for k in ("prep", "lstm_ih0", "lstm_layer#0", "lstm_ih1", "lstm_layer#1",
                                                   "lstm_outproj", "argvec", "mul_pl", "vis_enc", "obj_qkv", "obj_attn",
                                                   "obj_tail", "mul_pv", "mul_attn", "mul_tail", "pred_head")
                               if ktimes.get(k))),
                           "usec_per_step": (ktimes.get("lstm_layer#0") or lstm_us[1]) / T if not fused_ih else None,
                           "note": "(bytes: W_hh + W_ih + layer input + gates / state, each read once per launch) "
                                   "latency bound by design, not bandwidth bound: T dependent steps per launch, each "
                                   "one exchange of the Bn x R hidden vector among the 32 workgroups of a direction "
                                   "(self-validating 16-bit values in one slot per step, a 16-byte write-through per "
                                   "4 lanes and one 16-byte L1-bypassing load per thread: 1.3 us; 2.2 us per step with "
                                   "the LDS staging, 32 MFMAs per wave and the gates); before them the layer's input "
                                   "projection streams W_ih at the per-CU ingest rate (~40 GB/s from the Infinity "
                                   "Cache); W_hh is read once and stays in registers (64 CUs). "
                                   "In the forward it shares its launch with the encoders / the obj_tx tail "
                                   "(csrc/pair.hip), which run on the other CUs"}
        res["roofline_mfma"] = roof_mfma
    elif lstm_us and lstm_us[0] == "lstm_step" and 2 * T * lstm_us[1] > ktimes[mfma_dom]:
        nbytes = lstm_step_bytes(w, T)
        ach_b = nbytes / (lstm_us[1] * 1e-6) / 1e9
        res["roofline"] = {"bound": "hbm", "kernel": "lstm_step", "achieved": ach_b, "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": ach_b / PEAK_HBM_GBS,
                           "traffic": (pmc.get("kernels", {}).get("lstm_step") or {}).get("bytes_per_launch"),
                           "usec_per_launch": lstm_us[1], "bytes_per_launch": nbytes,
                           "launches_per_forward": 2 * T}
        res["roofline_mfma"] = roof_mfma
    pmc = pmc_traffic(args.workload + ("" if persistent else "_lstm_steps")) or {}
    # whole-forward fractions INSIDE `roofline` (the block the driver keeps): dense / executed MFMA work and the algorithmic bytes
    # (16-bit weights read once + the fp32 inputs + outputs) against the measured step time
    step_s = dt / args.steps
    alg_bytes = 88.4e6 + sum(v.numel() * v.element_size() for k, v in slots[0].inp.items()) + \
        sum(v.numel() * v.element_size() for k, v in slots[0].out.items() if torch.is_tensor(v))
    res["roofline"].update({
        "forward_frac_mfma_dense": total_flops / step_s / 1e12 / PEAK_MFMA_TFLOPS,
        "forward_frac_mfma_executed": executed_total / step_s / 1e12 / PEAK_MFMA_TFLOPS,
        "forward_frac_hbm_algorithmic": alg_bytes / step_s / 1e9 / PEAK_HBM_GBS,
        "forward_algorithmic_bytes": alg_bytes,
        "wasted_traffic_ratio": (pmc["bytes_per_forward"] / alg_bytes) if pmc.get("bytes_per_forward") else None})
    if pmc.get("bytes_per_forward"):
        # memory-side traffic by counters / the measured step time: a traffic RATE (it counts Infinity-Cache hits and the per-XCD
        # re-fetches of the weights: `wasted_traffic_ratio` above), not a roofline fraction
        gbs = pmc["bytes_per_forward"] / step_s / 1e9
        res["forward_traffic"] = {"bytes_per_forward_pmc": pmc["bytes_per_forward"], "rate_gbs": gbs, "source": pmc.get("source"),
                                  "note": "counter bytes / step time; the roofline fraction of the forward is "
                                          "roofline.forward_frac_hbm_algorithmic"}
    res["kernels_usec"] = {k: (round(v, 2) if v else None) for k, v in ktimes.items()}
    res["forward_roofline"] = {"algorithmic_gflop_per_batch": total_flops / 1e9,
                               "achieved_tflops": total_flops / (dt / args.steps) / 1e12,
                               "frac_of_mfma_peak": total_flops / (dt / args.steps) / 1e12 / PEAK_MFMA_TFLOPS,
                               "executed_gflop_per_batch": executed_total / 1e9,
                               "executed_tflops": executed_total / (dt / args.steps) / 1e12,
                               "frac_of_mfma_peak_executed": executed_total / (dt / args.steps) / 1e12 / PEAK_MFMA_TFLOPS,
                               "note": "algorithmic = the dense reference formulation (SURVEY.md 8(d)); executed = what the "
                                       "kernels compute: layer 0 of mul_tx projects visual and language rows once each "
                                       "(structured QKV) and attends nppf + nsrl instead of nsrl * nppf keys per query "
                                       "(separable attention, exact); MFMA tile padding is not counted as work"}
    if ktimes.get("mul_attn"):
        ka = (pmc.get("kernels", {}).get("mul_attn") or {})
        ach_x = executed_mul_attn / (ktimes["mul_attn"] * 1e-6) / 1e12
        ach_d = flops["mul_attn"] / (ktimes["mul_attn"] * 1e-6) / 1e12
        res["roofline_mul_attn"] = {"bound": "mfma", "kernel": "mul_attn (separable)", "usec_per_launch": ktimes["mul_attn"],
                                    "flops_executed": executed_mul_attn, "flops_dense_equivalent": flops["mul_attn"],
                                    "achieved": ach_x, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach_x / PEAK_MFMA_TFLOPS,
                                    "achieved_dense_equivalent": ach_d, "frac_dense_equivalent": ach_d / PEAK_MFMA_TFLOPS,
                                    "mfma_busy": ka.get("mfma_util_of_occupied_cus"),
                                    "note": "`frac` counts the FLOPs the kernel executes (separable softmax: nppf + nsrl instead of "
                                            "nsrl * nppf keys per query; at p100 also Q.K^T shared by the 5 arguments of a proposal); "
                                            "the dense-equivalent figure flatters a kernel that skips work and is quoted beside it only"}
    for blk in ("roofline", "roofline_mfma"):
        if blk in res and res[blk].get("kernel") in pmc.get("kernels", {}):
            res[blk]["mfma_busy"] = pmc["kernels"][res[blk]["kernel"]].get("mfma_util_of_occupied_cus")
    if world == 1 and w["conc"] in ("spat", "temp"):
        # the step in front of the forward: per-video items -> the slot's input tensors on the device
        # (vog_assemble_batch); together with the 63 GB/s host link this gives the PCIe-inclusive rate
        try:
            dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
            asm = dls.DeviceBatchAssembler(cfg, comm)
            it = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_items(w["B"], 4, nppf0, seed=3).items()}
            dst = {k: slots[0].inp[k] for k in dls.FWD_KEYS}
            for _ in range(5):
                asm(it, out=dst, with_loss_keys=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                asm(it, out=dst, with_loss_keys=False)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            nbytes = sum(it[k].numel() * 4 for k in dls.FWD_KEYS)
            h2d_us = nbytes / 63e9 * 1e6
            res["batch_assembly"] = {"usec_per_batch": us, "input_bytes": nbytes, "achieved_gbs": 2 * nbytes / us / 1e3,
                                     "pcie_h2d_usec_at_63GBs": h2d_us,
                                     "pcie_inclusive_queries_per_s": w["B"] / (max(h2d_us, dt / args.steps * 1e6) * 1e-6),
                                     "note": "raw per-video items cross PCIe once (8.7 MB per batch), the SPAT/TEMP layout is "
                                             "made on the device; with the copy overlapped the slower of (H2D, forward) bounds "
                                             "the rate - never reported as `value`"}
            if G == 1 and not args.no_graph and len(slots) >= 2:
                # the same K steps with the inputs STARTING IN HOST MEMORY: per step, on the slot's own stream, the
                # raw items go pinned host -> device, vog_assemble_batch writes the slot's input buffers, the
                # forward graph runs; copies of one slot overlap the forwards of the others
                ns = len(slots)
                # one packed pinned buffer per slot and ONE copy per step (dat_loader_simple.PackedStaging): the raw per-video
                # items the assembler reads + the word-level language arrays the forward reads
                lang_keys = ("srl_arg_words_ind", "srl_arg_word_mask", "srl_arg_word_mask_len", "srl_arg_words_capture",
                             "srl_arg_inds_msk", "num_cmp_msk")
                stg = [dls.PackedStaging({**{k: it[k].cpu() for k in dls.FWD_KEYS},
                                          **{k: slots[u].inp[k].cpu() for k in lang_keys}}, dev, n_dev=2) for u in range(ns)]
                # H2D copies on their own streams (they overlap the forwards of every slot); VOG_BENCH_COPY_STREAMS of them
                # (default 2: an 8.5 MB copy carries ~150 us of fixed latency on one stream - 27.7 GB/s back to back, 56 GB/s
                # for the 133 MB copies of cfg 4)
                ncs = max(1, int(os.environ.get("VOG_BENCH_COPY_STREAMS", "2")))
                copy_sts = [torch.cuda.Stream(device=dev) for _ in range(ncs)]
                dsts = [{k: slots[u].inp[k] for k in dls.FWD_KEYS} for u in range(ns)]
                sts = stream_pool[:ns] if len(stream_pool) >= ns else [torch.cuda.Stream(device=dev) for _ in range(ns)]

                zero_copy = os.environ.get("VOG_BENCH_ZERO_COPY", "0") == "1"

                def fed_step(i):
                    u = i % ns
                    if zero_copy:
                        with torch.cuda.stream(sts[u]):
                            asm({k: stg[u].host[k] for k in dls.FWD_KEYS}, out=dsts[u], with_loss_keys=False)
                            torch._foreach_copy_([slots[u].inp[k] for k in lang_keys], [stg[u].host[k] for k in lang_keys], non_blocking=True)
                            slots[u].launch(sts[u])
                        return
                    with torch.cuda.stream(sts[u]):
                        d = stg[u].upload_on(copy_sts[u % ncs])
                        asm({k: d[k] for k in dls.FWD_KEYS}, out=dsts[u], with_loss_keys=False)
                        torch._foreach_copy_([slots[u].inp[k] for k in lang_keys], [d[k] for k in lang_keys], non_blocking=True)
                        stg[u].release()
                        slots[u].launch(sts[u])

                fsteps = max(200, args.steps)
                for i in range(max(args.warmup, 2 * ns)):
                    fed_step(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(fsteps):
                    fed_step(i)
                torch.cuda.synchronize()
                dtf = time.perf_counter() - t0
                res["batch_assembly"]["measured_host_fed"] = {
                    "queries_per_s": w["B"] * fsteps / dtf, "us_per_step": dtf / fsteps * 1e6,
                    "h2d_bytes_per_step": stg[0].nbytes, "h2d_copies_per_step": 1,
                    "achieved_h2d_gbs": stg[0].nbytes * fsteps / dtf / 1e9,
                    "what": "same path, inputs in pinned host memory at the start of every step: ONE async H2D copy of a packed "
                            "staging buffer (raw per-video items + language arrays, dat_loader_simple.PackedStaging) on a copy "
                            "stream into one of two device buffers per slot, then device-side assembly + forward on the slot's "
                            "own stream; %d timed steps" % fsteps}
                if os.environ.get("VOG_BENCH_FED_GRAPH", "1") == "1":
                    # the same again with the feed INSIDE each slot's graph (engine.FedPipeline / Slot.feed_from): per step ONE
                    # transfer of the packed staging buffer (copy engine, copy stream) and ONE graph launch = device-side assembly +
                    # word-level arrays + forward; two fed slots per stream, one copy stream per forward stream.
                    # VOG_BENCH_FED_VIA = zero_copy | dma_node: the graph reads the pinned buffer itself / starts with a memcpy node
                    # (4 slots; both measured slower: profiles/round4_host_fed.md)
                    via = os.environ.get("VOG_BENCH_FED_VIA", "device")
                    fed_per_stream = max(1, int(os.environ.get("VOG_BENCH_FED_SLOTS_PER_STREAM", "2")))
                    spec = {k: v.clone() for k, v in stg[0].host.items()}
                    fed_bytes = stg[0].nbytes
                    if via == "device":
                        del stg
                        pipe = eng_mod.FedPipeline(eng, dict(slots[0].inp), spec, asm, streams=ns, slots_per_stream=fed_per_stream,
                                                      stream_pool=sts,
                                                      copy_streams={"own": "own", "old": copy_sts}.get(os.environ.get("VOG_BENCH_FED_COPY", ""), None))
                        nfs, fcs = len(pipe.slots), len(pipe.copy_streams)

                        def fedg_step(i):
                            pipe.submit()
                    else:
                        fstg = stg if via == "zero_copy" else [dls.PackedStaging(spec, dev, n_dev=1) for _ in range(ns)]
                        for u in range(ns):
                            slots[u].feed_from(fstg[u], asm, via=via)
                        nfs, fcs = ns, 0

                        def fedg_step(i):
                            slots[i % ns].launch(sts[i % ns])

                    for i in range(max(args.warmup, 2 * nfs)):
                        fedg_step(i)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(fsteps):
                        fedg_step(i)
                    torch.cuda.synchronize()
                    dtg = time.perf_counter() - t0
                    res["batch_assembly"]["measured_host_fed_graph"] = {
                        "queries_per_s": w["B"] * fsteps / dtg, "us_per_step": dtg / fsteps * 1e6,
                        "host_link_bytes_per_step": fed_bytes, "achieved_host_link_gbs": fed_bytes * fsteps / dtg / 1e9,
                        "form": via, "fed_slots": nfs, "copy_streams": fcs,
                        "what": "engine.FedPipeline: per step ONE copy-engine transfer of the packed pinned staging buffer (raw per-video "
                                "items + word-level arrays) and ONE graph launch whose first two kernels assemble the batch on the "
                                "device, then the forward; 2 fed slots per stream, a copy stream per forward stream; %d timed steps "
                                "(inputs in pinned host memory at the start of every step)" % fsteps}
        except Exception as e:          # never fail the bench line on the side measurement
            res["batch_assembly"] = {"error": str(e)}
    if world == 1 and not args.no_train_extra and cfg.mdl.name == "vog" and w["conc"] in ("temp", "spat") and not args.throughput_only:
        # side measurement, never `value`: the training step of SURVEY 8(f)-4 on the same workload (train.FP32Trainer: fp32
        # forward -> device loss -> backward of the whole network -> Adam), bounded to a few steps
        try:
            trn = importlib.import_module("vognet-pytorch_amd.train")
            sel = importlib.import_module("vognet-pytorch_amd.mdl_selector").get_mdl_loss_eval(cfg)
            loss_fn = sel["loss"](cfg, comm)
            tb = dict(batches[0])
            tb.update(synth.make_targets(tb, w["conc"], nppf0, seed=7))
            tdev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in tb.items()}
            tr = trn.FP32Trainer(cfg, comm, {k: torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v for k, v in sd.items()},
                                 loss_fn, lr=1e-4)
            l0 = float(tr.step(tdev)["loss"])
            tr.step(tdev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nt = 10
            for _ in range(nt):
                ld = tr.step(tdev)
            torch.cuda.synchronize()
            dtt = (time.perf_counter() - t0) / nt
            # the same with bf16-operand tile GEMMs (mixed precision option; fp32 accumulation and master weights)
            tr.bf16_gemm = True
            tr.step(tdev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nt):
                tr.step(tdev)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t0) / nt
            res["training_step"] = {"ms_per_step": dtt * 1e3, "queries_per_s": w["B"] / dtt, "dtype": "f32", "steps_timed": nt,
                                    "loss_first": l0, "loss_last": float(ld["loss"]), "ms_per_step_bf16_gemm": dtb * 1e3,
                                    "what": "train.FP32Trainer.step on the same batch: fp32 forward, vog_loss_fwd / _bwd, backward of both "
                                            "transformers, encoders, packed BiLSTM (BPTT), embedding, Adam (betas 0.9 / 0.99); pinned against "
                                            "autograd through the reference (tests/golden/bwd__*.npz)"}
            del tr
        except Exception as e:          # never fail the bench line on the side measurement
            res["training_step"] = {"error": str(e)}
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(w, cfg, sd, batches[0])
    print(json.dumps(res), file=out, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

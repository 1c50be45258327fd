"""Which HIP streams share a hardware queue? Throughput of the strict cfg-2 forward on subsets of 8 streams created up front
(2 streams were measured to be no faster than 1: the runtime maps streams onto its 4 hardware queues in some order)."""
import importlib, itertools, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
eng_mod = importlib.import_module("vognet-pytorch_amd.engine")
synth = importlib.import_module("vognet-pytorch_amd.synth")
ec = importlib.import_module("vognet-pytorch_amd.extended_config")
w = B.WORKLOADS["cfg2"]; cfg = B.make_cfg(w); nppf0 = ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = eng_mod.VogEngine(cfg, comm); eng.load_state_dict(synth.init_state_dict(cfg, B.VOCAB, seed=1))
NS = 8
streams = [torch.cuda.Stream() for _ in range(NS)]
slots = [eng.make_slot({k: torch.from_numpy(v) for k, v in synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=2000 + s).items()}, graph=True)
         for s in range(NS)]
def run(sub, K=400):
    eng_mod._LANE_BOOK.clear(); eng_mod._STREAM_LANE.clear()
    n = len(sub)
    for i in range(40): slots[sub[i % n]].launch(streams[sub[i % n]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): slots[sub[i % n]].launch(streams[sub[i % n]])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for sub in [(0,), (0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (0, 1, 2), (0, 1, 2, 3), (0, 2, 4, 6), (1, 3, 5, 7), (0, 1, 4, 5), (2, 3, 6, 7), (4, 5, 6, 7), (1, 2, 3, 4)]:
    us = run(sub)
    print(f"streams {sub}: {us:.1f} us per batch = {4e6 / us:.0f} queries/s")

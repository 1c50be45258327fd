import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench as B
eng_mod = importlib.import_module("vognet-pytorch_amd.engine"); synth = importlib.import_module("vognet-pytorch_amd.synth"); ec = importlib.import_module("vognet-pytorch_amd.extended_config")
w = B.WORKLOADS["cfg2"]; cfg = B.make_cfg(w); nppf0 = ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = eng_mod.VogEngine(cfg, comm); eng.load_state_dict(synth.init_state_dict(cfg, B.VOCAB, seed=1))
def run(ns, create_first):
    eng_mod._LANE_BOOK.clear(); eng_mod._STREAM_LANE.clear()
    streams, slots = [], []
    if create_first:
        streams = [torch.cuda.Stream() for _ in range(ns)]
    for s in range(ns):
        if not create_first:
            streams.append(torch.cuda.Stream())
        slots.append(eng.make_slot({k: torch.from_numpy(v) for k, v in synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=2000 + s).items()}, graph=True))
    for i in range(40): slots[i % ns].launch(streams[i % ns])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): slots[i % ns].launch(streams[i % ns])
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 400 * 1e6
    print(f"{ns} streams, streams created {'first' if create_first else 'interleaved with slots'}: {us:.1f} us/batch; stream ids {[hex(s.cuda_stream)[-5:] for s in streams]}")
for ns in (2, 3, 4):
    run(ns, True); run(ns, False)


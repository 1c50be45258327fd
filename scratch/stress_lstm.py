"""Stress the persistent BiLSTM hand-off: 4 (or more) graphs in flight for many iterations; every
output must stay bit-identical to the slot's first result (the kernel's arithmetic does not depend
on timing; a stale or torn hand-off word would show up as a difference or a NaN)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
ec, synth, eng_mod = bench.ec, bench.synth, bench.eng_mod
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
nslots = int(sys.argv[3]) if len(sys.argv) > 3 else 4
w = bench.WORKLOADS[wl]
cfg = bench.make_cfg(w)
nppf0 = ec.num_prop_per_frm(cfg)
comm = {"vocab_size": bench.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = eng_mod.VogEngine(cfg, comm)
eng.load_state_dict(synth.init_state_dict(cfg, bench.VOCAB, seed=1))
slots, streams = [], []
for s in range(nslots):
    b = synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=bench.VOCAB, seed=77 + s, ragged=True)
    slots.append(eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, graph=True))
    streams.append(torch.cuda.Stream())
refs = []
for sl, st in zip(slots, streams):
    sl.launch(st)
torch.cuda.synchronize()
for sl in slots:
    refs.append({k: v.clone() for k, v in sl.out.items() if isinstance(v, torch.Tensor)})
    assert torch.isfinite(sl.out["mdl_outs"]).all()
t0 = time.time()
bad = 0
for i in range(iters):
    s = i % nslots
    slots[s].launch(streams[s])
    if (i + 1) % 400 == 0:
        torch.cuda.synchronize()
        for sl, ref in zip(slots, refs):
            for k in ref:
                if not torch.equal(sl.out[k], ref[k]):
                    bad += 1
                    print("MISMATCH at iteration", i, k, (sl.out[k] - ref[k]).abs().max().item())
        if bad:
            break
torch.cuda.synchronize()
print(f"{wl}: {iters} launches on {nslots} streams in {time.time() - t0:.1f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)

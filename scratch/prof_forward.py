"""N eager forwards of one workload on one stream (profiling driver: no timing loops, no graphs).
usage: python scratch/prof_forward.py <cfg2|cfg3|cfg4|cfg5> <N> [pair=0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
wl, N = sys.argv[1], int(sys.argv[2])
pair = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = B.WORKLOADS[wl]
cfg = B.make_cfg(w)
nppf0 = B.ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = B.eng_mod.VogEngine(cfg, comm)
eng.load_state_dict(B.synth.init_state_dict(cfg, B.VOCAB, seed=1))
eng.set_option("pair_launches", pair)
b = B.synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=7)
slot = eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, graph=False)
for _ in range(N):
    slot.launch()
torch.cuda.synchronize()
print("forwards", N, "checksum", float(slot.out["mdl_outs_eval"].sum()))

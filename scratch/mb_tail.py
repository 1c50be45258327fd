"""Per-kernel HIP-event times of the fused launches of one cfg-2 forward (vog_time_kernel).
usage: [VOG_PERF_EXPERIMENTS=1 VOG_TAIL_DEBUG=k] python scratch/mb_tail.py [kernel ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
w = B.WORKLOADS[os.environ.get("WL", "cfg2")]
w = dict(w, tx=os.environ.get("TX", w["tx"]))
cfg = B.make_cfg(w)
nppf0 = B.ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = B.eng_mod.VogEngine(cfg, comm)
eng.load_state_dict(B.synth.init_state_dict(cfg, B.VOCAB, seed=1))
for kv in os.environ.get("OPTS", "").split():          # engine switches, e.g. OPTS="qkv_lean=0 pair_launches=0"
    k, v = kv.split("=")
    eng.set_option(k, int(v))
b = B.synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=7)
slot = eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, graph=False)
slot.launch(); torch.cuda.synchronize()
names = sys.argv[1:] or ["obj_tail", "mul_tail", "vis_enc"]
out = []
for n in names:
    try:
        out.append(f"{n} {eng.time_kernel(slot, n, 200):.2f}")
    except Exception as e:
        out.append(f"{n} n/a")
print(os.environ.get("VOG_TAIL_DEBUG", "0"), " ".join(out))

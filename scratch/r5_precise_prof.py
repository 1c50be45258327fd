"""fp32 (precise) forward of cfg 2: time per forward; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
w = dict(B.WORKLOADS[os.environ.get("WL", "cfg2")], tx="f32")
cfg = B.make_cfg(w)
nppf0 = B.ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = B.eng_mod.VogEngine(cfg, comm)
eng.load_state_dict(B.synth.init_state_dict(cfg, B.VOCAB, seed=1))
assert eng.precise is not None
b = B.synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=7)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
for _ in range(3):
    eng.forward(dev)
torch.cuda.synchronize()
N = int(os.environ.get("N", "20"))
t0 = time.perf_counter()
for _ in range(N):
    eng.forward(dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"precise forward: {dt * 1e3:.3f} ms per batch = {w['B'] / dt:.0f} queries/s")

"""HIP-event timing of vog_mul_tail_bwd at the cfg-2 shape (M = 4000 rows, d = 768)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
bwd = importlib.import_module("vognet-pytorch_amd.backward")
synth = importlib.import_module("vognet-pytorch_amd.synth")
import bench as B
cfg = B.make_cfg(B.WORKLOADS["cfg2"])
sd = {k: torch.from_numpy(v) for k, v in synth.init_state_dict(cfg, B.VOCAB, seed=1).items()}
M, d = 4000, 768
attn = torch.randn(M, d, device="cuda"); x = torch.randn(M, d, device="cuda"); dm = torch.randn(4, 5, 200, device="cuda") * 1e-3
for _ in range(3): bwd.mul_tail_backward(sd, 0, attn, x, dm, 4, 10, 20, 5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
sdd = {k: v.cuda() for k, v in sd.items()}
e0.record()
for _ in range(10): bwd.mul_tail_backward(sdd, 0, attn, x, dm, 4, 10, 20, 5)
e1.record(); torch.cuda.synchronize()
print("vog_mul_tail_bwd cfg2: %.1f us per call (33 GFLOP incl. the fp32 recomputation)" % (e0.elapsed_time(e1) * 100))

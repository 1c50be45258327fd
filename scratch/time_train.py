"""Time of one device training step (train.FP32Trainer.step) at BASELINE config 2; gpurun_out/train_time.json."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases
synth = importlib.import_module("vognet-pytorch_amd.synth")
trn = importlib.import_module("vognet-pytorch_amd.train")
sel_mod = importlib.import_module("vognet-pytorch_amd.mdl_selector")
name = "full/cfg2_vog_spat_gt5_bs4"
cfg, sd, batch, c = cases.build(name)
comm = {"vocab_size": c["vocab"], "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": c["nppf0"]}
sel = sel_mod.get_mdl_loss_eval(cfg)
loss_fn = sel["loss"](cfg, comm)
tg = synth.make_targets(batch, cfg.ds.conc_type, c["nppf0"], seed=c["dseed"])
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in {**batch, **tg}.items()}
BF16 = bool(int(os.environ.get("TRAIN_BF16", "0")))
tr = trn.FP32Trainer(cfg, comm, {k: torch.from_numpy(v) for k, v in sd.items()}, loss_fn, lr=1e-4, bf16_gemm=BF16,
                     dropout=bool(int(os.environ.get("TRAIN_DROPOUT", "0"))))
for _ in range(3):
    tr.step(dev)
torch.cuda.synchronize()
t0 = time.time(); n = 20
for _ in range(n):
    ld = tr.step(dev)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
t1 = time.time()
for _ in range(n):
    tr.forward(dev)
torch.cuda.synchronize()
df = (time.time() - t1) / n
res = {"config": name, "bf16_gemm": BF16, "ms_per_train_step": dt * 1e3, "ms_fp32_forward": df * 1e3, "queries_per_s_training": 4 / dt, "loss_after": float(ld["loss"]),
       "parameters": int(sum(v.numel() for v in tr.params.values()))}
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "train_time.json"), "w"))

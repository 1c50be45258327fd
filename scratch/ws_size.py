import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch
import bench
ec, synth, eng_mod = bench.ec, bench.synth, bench.eng_mod
w = bench.WORKLOADS["cfg2"]
cfg = bench.make_cfg(w)
comm = {"vocab_size": bench.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": ec.num_prop_per_frm(cfg)}
eng = eng_mod.VogEngine(cfg, comm)
eng.load_state_dict(synth.init_state_dict(cfg, bench.VOCAB, seed=1))
print("workspace MB", eng.lib.vog_workspace_bytes(eng.ctx, 4, 4, 12) / 1e6)
import ctypes as C
L = importlib.import_module("vognet-pytorch_amd.lib")
tot = 0
for st in ("prop16","seg16","enc_slabs","prop_seg","xmul","xmul16","mul_q","mul_k","mul_vt","mul_attn16","mul_tmp","mul_x1","mul_x1_16","mul_ffn16","mul_outA","mul_outA16","mul_pv","h1","gx","obj_q","obj_tmp"):
    off, nb = C.c_int64(), C.c_int64()
    if eng.lib.vog_workspace_stage(eng.ctx, 4, 4, 12, st.encode(), C.byref(off), C.byref(nb)) == 0:
        print(f"{st:12s} {nb.value/1e6:8.2f} MB")

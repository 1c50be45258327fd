#!/bin/bash
# mul tail (cfg 2: 63 workgroups; cfg 4: 1250) under the VOG_TAIL_DEBUG ablations: 1 = every weight load hits the block's first KiB,
# 2 = no matrix work, 4 = no GEMM stage at all (skeleton). Results are wrong by construction; only the kernel time is read.
for w in cfg2 cfg4; do for d in 0 1 2 3 4; do
  echo -n "$w VOG_TAIL_DEBUG=$d: "
  VOG_PERF_EXPERIMENTS=1 VOG_TAIL_DEBUG=$d python - <<PY 2>/dev/null
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
import bench as B
ec, synth, eng_mod = B.ec, B.synth, B.eng_mod
w = B.WORKLOADS["$w"]; cfg = B.make_cfg(w); nppf0 = ec.num_prop_per_frm(cfg)
comm = {"vocab_size": B.VOCAB, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": nppf0}
eng = eng_mod.VogEngine(cfg, comm); eng.load_state_dict(synth.init_state_dict(cfg, B.VOCAB, seed=1))
b = synth.make_batch(w["conc"], w["B"], nppf0, vocab_size=B.VOCAB, seed=5)
slot = eng.make_slot({k: torch.from_numpy(v) for k, v in b.items()}, graph=False)
slot.launch(); torch.cuda.synchronize()
print("mul_tail %.1f us  obj_tail %.1f us" % (eng.time_kernel(slot, "mul_tail", 50), eng.time_kernel(slot, "obj_tail", 50)))
PY
done; done

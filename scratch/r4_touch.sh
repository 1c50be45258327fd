#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -k "tail or golden or pair" 2>&1 | tail -3
for w in cfg2 cfg4; do
  echo -n "$w: "
  python - <<PY 2>/dev/null
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
done
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2 3; do
echo -n "cfg2 4 streams: "; python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "cfg2 1 stream:  "; python bench.py $A --steps 1000 --warmup 40 --streams 1 2>/dev/null | tail -1
echo -n "cfg4:           "; python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done
echo -n "cfg3: "; python bench.py $A --workload cfg3 --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "cfg5: "; python bench.py $A --workload cfg5 --steps 2000 --warmup 40 2>/dev/null | tail -1

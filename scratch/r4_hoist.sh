#!/bin/bash
# tail: first weight loads of every stage hoisted in front of the barrier-bound work before it (VOG_TAIL_HOIST bit mask)
for v in base h7 h23 h31; do L=""; [ $v != base ] && L=/root/repo/scratch/tmp/$v/libvog_hip.so
for w in cfg2 cfg4; do
  echo -n "$v $w: "
  VOG_HIP_LIB=$L python - <<PY 2>/dev/null
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
print("mul_tail %.1f us  obj_tail %.1f us" % (eng.time_kernel(slot, "mul_tail", 100), eng.time_kernel(slot, "obj_tail", 100)))
PY
done
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do
echo -n "$v cfg2 4 streams: "; VOG_HIP_LIB=$L python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "$v cfg2 1 stream:  "; VOG_HIP_LIB=$L python bench.py $A --steps 1000 --warmup 40 --streams 1 2>/dev/null | tail -1
done
echo -n "$v cfg4: "; VOG_HIP_LIB=$L python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done
VOG_HIP_LIB=/root/repo/scratch/tmp/h23/libvog_hip.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -k "tail or golden" 2>&1 | tail -2

#!/bin/bash
# LSTM layer kernel at <= 192 registers (B-fragment groups of 2 / 4 instead of 8): does a lean workgroup of another stream share its CUs?
O=gpurun_out/r4lstmg; mkdir -p $O
A="--steps 3000 --warmup 100 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for v in base g2 g4; do
  L=""; [ $v != base ] && L=/root/repo/scratch/tmp/$v/libvog_hip.so
  for rep in 1 2; do
    VOG_HIP_LIB=$L python bench.py $A > $O/$v.$rep.json 2> $O/$v.$rep.err
    VOG_HIP_LIB=$L python bench.py $A --streams 1 --steps 1000 > $O/$v.s1.$rep.json 2>> $O/$v.$rep.err
  done
done
python - <<'PY'
import json
for v in ("base", "g2", "g4"):
    for rep in (1, 2):
        d = json.loads(open(f"gpurun_out/r4lstmg/{v}.{rep}.json").read().strip().splitlines()[-1])
        s = json.loads(open(f"gpurun_out/r4lstmg/{v}.s1.{rep}.json").read().strip().splitlines()[-1])
        print(v, rep, "4 streams", round(d["value"]), d["parity"]["ok"], "1 stream", round(s["value"]), s["ms_per_step"], [ (k, d[k]) for k in d if "lstm" in k.lower()][:3])
PY

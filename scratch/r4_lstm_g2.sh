#!/bin/bash
# does a lean workgroup really share a CU with the <= 192-register BiLSTM layer kernel? 64 / 128 CUs (HSA_CU_MASK): the layer's 64 workgroups
# hold every CU of a 64-CU chip, so other streams progress during a layer only by co-residency
O=gpurun_out/r4lstmg2; mkdir -p $O
A="--steps 1500 --warmup 100 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for v in base g2; do
  L=""; [ $v != base ] && L=/root/repo/scratch/tmp/$v/libvog_hip.so
  for m in 63 127; do for s in 1 2 4; do
    HSA_CU_MASK=0:0-$m VOG_HIP_LIB=$L timeout 120 python bench.py $A --streams $s > $O/$v.$m.$s.json 2> $O/$v.$m.$s.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/$v.$m.$s.json").read().strip().splitlines()[-1]); print("$v cus", $m + 1, "streams", $s, round(d["value"]), d["parity"]["ok"])
except Exception as e: print("$v $m $s failed", e)
PY
  done; done
done

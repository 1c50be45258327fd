#!/bin/bash
# host-fed loop: DMA copy of the packed staging buffer vs the assembler reading pinned host memory directly (zero copy)
O=gpurun_out/r4zc; mkdir -p $O
A="--steps 400 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for z in 0 1; do for w in cfg2 cfg4; do
  VOG_BENCH_ZERO_COPY=$z timeout 300 python bench.py $A --workload $w > $O/z$z.$w.json 2> $O/z$z.$w.err
  python - <<PY
import json
d = json.loads(open("$O/z$z.$w.json").read().strip().splitlines()[-1])
print("zero_copy=$z $w value", round(d["value"]), "host fed", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["batch_assembly"].get("measured_host_fed", d["batch_assembly"]).items() if k != "what"})
PY
done; done

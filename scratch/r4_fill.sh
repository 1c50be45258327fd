#!/bin/bash
# T(K) from a drained pipeline: the driver's --steps 20 --warmup 5 against longer runs
O=gpurun_out/r4fill; mkdir -p $O
for K in 4 8 12 16 20 24 40 80 400; do for r in 1 2 3; do
  python bench.py --steps $K --warmup 5 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only > $O/k$K.$r.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/k$K.$r.json").read().strip().splitlines()[-1])
print("K=$K run $r: value", round(d["value"]), "T_total_us", round(d["ms_per_step"] * $K * 1e3, 1))
PY
done; done

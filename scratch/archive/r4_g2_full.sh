#!/bin/bash
O=gpurun_out/r4g2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --workload $w --no-train-extra --no-cpu-baseline > $O/$w.json 2> $O/$w.err
  python - <<PY
import json
d = json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
print("$w value", round(d["value"]), d["parity"]["ok"], "hbm inputs", round(d.get("value_hbm_inputs") or 0), "fed graph", round(d.get("batch_assembly", {}).get("measured_host_fed_graph", {}).get("queries_per_s", 0)))
PY
done

#!/bin/bash
O=gpurun_out/r4fed5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_surface.py -x -q -k "fed_ or copy_segments or prefetcher or evaluator or batch_requests" 2>&1 | tail -5 > $O/tests.log
for w in cfg2 cfg3 cfg4; do
  timeout 400 python bench.py --workload $w --no-train-extra > $O/$w.json 2> $O/$w.err
  python - <<PY
import json
d = json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$w value", round(d["value"]), d["parity"]["ok"], "hbm inputs", round(d.get("value_hbm_inputs") or 0))
for k in ("measured_host_fed", "measured_host_fed_graph"):
    print("   ", k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in ba.get(k, {"missing": ba.get("error")}).items() if a != "what"})
PY
done

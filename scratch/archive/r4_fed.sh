#!/bin/bash
O=gpurun_out/r4fed; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_forward.py -x -q -k "fed_slot or copy_segments or slot_replay" 2>&1 | tail -15 > $O/tests.log
A="--steps 400 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for w in cfg2 cfg4 cfg3; do
  timeout 300 python bench.py $A --workload $w > $O/$w.json 2> $O/$w.err
  python - <<PY
import json
d = json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$w value", round(d["value"]), d["parity"]["ok"])
for k in ("measured_host_fed", "measured_host_fed_graph"):
    print("   ", k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in ba.get(k, {"missing": ba.get("error")}).items() if a != "what"})
PY
done

#!/bin/bash
O=gpurun_out/r4fed4; mkdir -p $O
A="--steps 400 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for w in cfg2 cfg3 cfg4; do for sps in 1 2 3; do for ncs in 2 4; do
  [ $w = cfg4 ] && [ $sps = 3 ] && continue
  VOG_BENCH_COPY_STREAMS=$ncs VOG_BENCH_FED_VIA=device VOG_BENCH_FED_SLOTS_PER_STREAM=$sps timeout 300 python bench.py $A --workload $w > $O/$w.$sps.$ncs.json 2> $O/$w.$sps.$ncs.err
  python - <<PY
import json
d = json.loads(open("$O/$w.$sps.$ncs.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$w slots/stream=$sps copy streams=$ncs value", round(d["value"]), d["parity"]["ok"], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in ba.get("measured_host_fed_graph", {"missing": ba.get("error")}).items() if a != "what"})
PY
done; done; done

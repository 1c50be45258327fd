#!/bin/bash
O=gpurun_out/r4fed6; mkdir -p $O
run() { n=$1; shift; timeout 400 python bench.py --workload cfg2 --no-train-extra "$@" > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
d = json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$n", "$*", "value", round(d["value"]), "fed", round(ba["measured_host_fed"]["queries_per_s"]), "fed graph", round(ba["measured_host_fed_graph"]["queries_per_s"]))
PY
}
run a --no-cpu-baseline
run b --rotate-inputs 0
run c --no-cpu-baseline --rotate-inputs 0
run d --no-cpu-baseline --rotate-inputs 0 --no-cobatch-extra
run e

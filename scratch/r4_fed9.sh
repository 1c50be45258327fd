#!/bin/bash
O=gpurun_out/r4fed9; mkdir -p $O
run() { n=$1; shift; timeout 400 python bench.py --workload ${W:-cfg2} --no-train-extra --no-cpu-baseline --rotate-inputs 0 --no-cobatch-extra > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
d = json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$n", "value", round(d["value"]), "fed", round(ba["measured_host_fed"]["queries_per_s"]), "fed graph", round(ba["measured_host_fed_graph"]["queries_per_s"]), ba["measured_host_fed_graph"]["fed_slots"], ba["measured_host_fed_graph"]["copy_streams"])
PY
}
for r in 1 2 3; do VOG_BENCH_COPY_STREAMS=2 run p2.$r; done
for r in 1 2 3; do VOG_BENCH_COPY_STREAMS=1 run p1.$r; done
for r in 1 2 3; do VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=2 run own2.$r; done

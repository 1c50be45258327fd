#!/bin/bash
O=gpurun_out/r4fed10; mkdir -p $O
run() { n=$1; shift; timeout 400 python bench.py --workload ${W:-cfg2} --no-train-extra --no-cpu-baseline --rotate-inputs 0 --no-cobatch-extra > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
d = json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$n", "numa", d.get("host_numa_node"), "value", round(d["value"]), "fed", round(ba["measured_host_fed"]["queries_per_s"]), "fed graph", round(ba["measured_host_fed_graph"]["queries_per_s"]), ba["measured_host_fed_graph"]["fed_slots"], ba["measured_host_fed_graph"]["copy_streams"])
PY
}
for r in 1 2 3 4; do run bind.$r; done
for r in 1 2 3 4; do VOG_BENCH_NUMA_BIND=0 run nobind.$r; done
for r in 1 2; do VOG_BENCH_FED_COPY=own run own.$r; done

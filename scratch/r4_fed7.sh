#!/bin/bash
O=gpurun_out/r4fed7; mkdir -p $O
run() { n=$1; shift; timeout 400 python bench.py --workload ${W:-cfg2} --no-train-extra --no-cpu-baseline --rotate-inputs 0 --no-cobatch-extra > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
d = json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$n", "value", round(d["value"]), "fed", round(ba["measured_host_fed"]["queries_per_s"]), "fed graph", round(ba["measured_host_fed_graph"]["queries_per_s"]), ba["measured_host_fed_graph"]["fed_slots"], ba["measured_host_fed_graph"]["copy_streams"])
PY
}
VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=1 run own1
VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=2 run own2
VOG_BENCH_COPY_STREAMS=4 VOG_BENCH_FED_COPY=old VOG_BENCH_FED_SLOTS_PER_STREAM=2 run old4_2
VOG_BENCH_COPY_STREAMS=2 VOG_BENCH_FED_COPY=old VOG_BENCH_FED_SLOTS_PER_STREAM=2 run old2_2
VOG_BENCH_COPY_STREAMS=4 VOG_BENCH_FED_SLOTS_PER_STREAM=2 run new_after4
VOG_BENCH_COPY_STREAMS=2 VOG_BENCH_FED_SLOTS_PER_STREAM=2 run new_after2
W=cfg3 VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=1 run c3own1
W=cfg3 VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=2 run c3own2
W=cfg4 VOG_BENCH_FED_COPY=own VOG_BENCH_FED_SLOTS_PER_STREAM=1 run c4own1

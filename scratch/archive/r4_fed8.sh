#!/bin/bash
O=gpurun_out/r4fed8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -k "fed_" 2>&1 | tail -5 > $O/tests.log
run() { n=$1; shift; timeout 400 python bench.py --workload ${W:-cfg2} --no-train-extra --no-cpu-baseline --rotate-inputs 0 --no-cobatch-extra > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
d = json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$n", "value", round(d["value"]), "fed", round(ba["measured_host_fed"]["queries_per_s"]), "fed graph", round(ba["measured_host_fed_graph"]["queries_per_s"]), ba["measured_host_fed_graph"]["fed_slots"], ba["measured_host_fed_graph"]["copy_streams"])
PY
}
VOG_BENCH_COPY_STREAMS=2 run p2
VOG_BENCH_COPY_STREAMS=3 run p3
VOG_BENCH_COPY_STREAMS=4 run p4
VOG_BENCH_COPY_STREAMS=1 run p1
VOG_BENCH_COPY_STREAMS=2 VOG_BENCH_FED_SLOTS_PER_STREAM=3 run p2s3
W=cfg3 run c3
W=cfg4 run c4
